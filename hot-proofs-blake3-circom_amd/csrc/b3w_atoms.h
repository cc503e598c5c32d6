// b3w_atoms.h — "atoms" (value-level names of circuit signals), their numbering, and where the
// trace phase of the kernels parks each atom in LDS.  Shared by host (slot-table construction) and
// device (trace phase).  Numbering is the contract of hot-proofs-blake3-circom_amd/layouts/*.layout.
//
// An atom is the value of one signal of the reference circuits, named after the template that
// produces it (reference paths relative to the reference repo):
//   H, M, T, B, D, O      inputs / outputs of Blake3Compression   circuits/blake3_compression.circom:171-228
//   half-G atoms          HalfFunG(a,b,c,d,R1,R2)                  circuits/blake3_compression.circom:72-100
//       S1 = add1.inp (Bits34, blake3_common.circom:183-203), A = add1.out_word,
//       S3 = add3.inp (Bits33, :160-178),                       C = add3.out_word,
//       D2 = rxor2.out_word, DI = rxor2.tb.inp (RotXorWordBits, blake3_compression.circom:53-67),
//       B4 = rxor4.out_word, BI = rxor4.tb.inp
//   nova atoms            Blake3Nova and helpers                    circuits/blake3_nova.circom:13-267
#pragma once
#include <stdint.h>

#define B3W_A_ONE 0
#define B3W_A_H 1
#define B3W_A_M 9
#define B3W_A_T 25
#define B3W_A_B 27
#define B3W_A_D 28
#define B3W_A_O 29
#define B3W_A_HG 45     // + 8*k + j, k = (round*8 + g)*2 + half, j below
#define B3W_A_NV 941    // nova narrow atoms
#define B3W_HG_S1 0
#define B3W_HG_A 1
#define B3W_HG_S3 2
#define B3W_HG_C 3
#define B3W_HG_D2 4
#define B3W_HG_DI 5
#define B3W_HG_B4 6
#define B3W_HG_BI 7

// nova narrow atoms: offsets from B3W_A_NV
enum {
  NV_N_BLOCKS = 0, NV_BLOCK_COUNT = 1, NV_H = 2, NV_CIL = 10, NV_CIH = 11, NV_LEAF_DEPTH = 12,
  NV_TOTAL_DEPTH = 13, NV_DEPTH = 14, NV_M = 15, NV_B = 31, NV_BLOCK_COUNT_OUT = 32, NV_DEPTH_OUT = 33,
  NV_IS_ROOT = 34, NV_IS_PARENT = 35, NV_CP_IN1 = 36, NV_CP_N2B_IN = 37, NV_ED_IN1 = 38, NV_ED_N2B_IN = 39,
  NV_ED_OUT = 40, NV_NOT_ROOT = 41, NV_NOT_PARENT = 42, NV_E0 = 43, NV_E1 = 44, NV_IS_LAST_BLOCK = 45,
  NV_FIRST = 46, NV_UR_TMP = 47, NV_UR_FLAG = 48, NV_CHUNK_IDX = 49, NV_DL = 50, NV_CDD_OUT = 51,
  NV_DECR_DEPTH = 52, NV_TMP_DOWN = 53, NV_M_IS_PARENT = 69, NV_TMP_IS_PAR = 85, NV_TMPIV = 101,
  NV_EQ_OUT = 109, NV_BIT_AT_DEPTH = 173, NV_NARROW_COUNT = 237,
  // wide atoms (full field elements), continuing the numbering
  NV_ROOT_INV = 237, NV_E0_INV = 238, NV_E1_INV = 239, NV_ROOT_ISZ_IN = 240, NV_E0_ISZ_IN = 241,
  NV_E1_ISZ_IN = 242, NV_E1_IN1 = 243, NV_EQ_INV = 244, NV_EQ_ISZ_IN = 308, NV_EQ_IN1 = 372, NV_COUNT = 436
};
#define B3W_N_COMP_ATOMS 941
#define B3W_N_NOVA_ATOMS (B3W_A_NV + NV_COUNT)

// ---- LDS image of one witness' trace (u32 words) -------------------------------------------
//   [0,45)            atoms 0..44 (ONE H M T B D O), one word each
//   [48 + 8k, +8)     half-G k: S1.lo S1.hi S3.lo S3.hi D2 DI B4 BI   (A == S1.lo, C == S3.lo)
//   [944, 944+237)    nova narrow atoms, one word each
//   [1182, 1184)      chunk_idx (64-bit)
//   [1184 + 8j, +8)   wide atom j (256-bit).  O2 builds keep only the 67 IsZero inverses:
//                     j = 0..2 root/e0/e1 inv, 3..66 eq_inv[0..63];   O1 keeps all 199 in
//                     numbering order.
#define B3W_LDS_OKWORD 45   // unused pad word: 1 = witness valid (two-kernel path)
#define B3W_LDS_HG 48
#define B3W_LDS_NV 944
#define B3W_LDS_CHUNK_IDX 1182
#define B3W_LDS_WIDE 1184
#define B3W_LDS_WORDS_COMP 944
#define B3W_LDS_WORDS_NOVA_O2 (1184 + 8 * 67)
#define B3W_LDS_WORDS_NOVA_O1 (1184 + 8 * 199)

// slot-table entry: what the expand phase stores into one 32-byte witness slot
//   bits 0..11 src word in the LDS image, bits 12..16 shift, bits 17..18 mode
#define B3W_MODE_BIT 0u   // (lds[src] >> sh) & 1
#define B3W_MODE_W32 1u   // lds[src]
#define B3W_MODE_W64 2u   // lds[src], lds[src+1]
#define B3W_MODE_W256 3u  // lds[src .. src+7]
#define B3W_ENTRY(src, sh, mode) ((uint32_t)(src) | ((uint32_t)(sh) << 12) | ((uint32_t)(mode) << 17))

// circuit "kinds" the kernels are specialised on
#define B3W_KIND_COMP 0
#define B3W_KIND_NOVA_O2 1
#define B3W_KIND_NOVA_O1 2
