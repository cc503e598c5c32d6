/*
 * b3w_oracle.c — CPU restatement of the reference's witness computation for the BLAKE3 circom
 * circuits.  TEST INFRASTRUCTURE ONLY: linked/loaded only by tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg.  The product (hot-proofs-blake3-circom_amd/csrc) never calls it.
 *
 * Parity pinning: checked in tests/test_oracle_golden.py against
 *   - the reference's committed golden witness build/blake3_compression/testInp/witness.wtns
 *     (sha256 0c3f9a39...606f; committed gz copy under tests/golden/),
 *   - fixtures generated in the build container by running the reference's own committed WASM
 *     through its witness_calculator.js (tools/gen_golden.py -> tests/golden/ *.json).
 *
 * Every signal is evaluated as a full field element in [0,p) with circom semantics
 * (`>>` and `&` act on the canonical representative), so non-canonical inputs (negative,
 * >= 2^32) behave as in the WASM, including "Assert Failed".
 *
 * Follows (relative to /root/reference):
 *   circuits/blake3_common.circom:15-26    Blake3Permute (sigma)
 *   circuits/blake3_common.circom:142-154  ToBits   :160-178 Bits33   :183-203 Bits34
 *   circuits/blake3_common.circom:55-80    XorWord2
 *   circuits/blake3_compression.circom:17-24 IV  :29-67 RotXor[Word]Bits  :72-100 HalfFunG
 *   circuits/blake3_compression.circom:106-123 MixFunG  :128-161 SingleRound  :171-228 Blake3Compression
 *   circuits/blake3_nova.circom:13-45 CheckDepth  :47-84 GetDownLeftPath  :86-120 GetFinal_m
 *   circuits/blake3_nova.circom:122-167 GetFlag   :169-267 Blake3Nova
 *   blake3_nova_js/witness_calculator.js:208-272 .wtns image
 *   circomlib 2.0.5 (yarn.lock:1243; not vendored): IsZero/IsEqual/LessThan/GreaterEqThan/
 *   Num2Bits/NOT/AND/OR restated from the published templates.
 *
 * Atom numbering and layout files: see tools/b3w_model.py and
 * hot-proofs-blake3-circom_amd/layouts/ *.layout (slot -> atom / bit-of-atom).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t l[4]; } fe;

typedef struct {
  fe p;          /* modulus */
  fe r2;         /* R^2 mod p, R = 2^256 */
  uint64_t n0;   /* -p^-1 mod 2^64 */
} field_t;

enum { B3WO_COMPRESSION = 0, B3WO_NOVA_BN254 = 1, B3WO_NOVA_VESTA = 2, B3WO_NOVA_BN254_O1 = 3, B3WO_NCIRCUITS = 4 };

enum { A_ONE = 0, A_H = 1, A_M = 9, A_T = 25, A_B = 27, A_D = 28, A_O = 29, A_HG = 45, A_NV = 941 };
enum { HG_S1, HG_A, HG_S3, HG_C, HG_D2, HG_DI, HG_B4, HG_BI };
/* nova narrow atoms, offsets from A_NV (order = tools/b3w_model.py NOVA_NARROW) */
enum {
  NV_N_BLOCKS = 0, NV_BLOCK_COUNT = 1, NV_H = 2, NV_CIL = 10, NV_CIH = 11, NV_LEAF_DEPTH = 12,
  NV_TOTAL_DEPTH = 13, NV_DEPTH = 14, NV_M = 15, NV_B = 31, NV_BLOCK_COUNT_OUT = 32, NV_DEPTH_OUT = 33,
  NV_IS_ROOT = 34, NV_IS_PARENT = 35, NV_CP_IN1 = 36, NV_CP_N2B_IN = 37, NV_ED_IN1 = 38, NV_ED_N2B_IN = 39,
  NV_ED_OUT = 40, NV_NOT_ROOT = 41, NV_NOT_PARENT = 42, NV_E0 = 43, NV_E1 = 44, NV_IS_LAST_BLOCK = 45,
  NV_FIRST = 46, NV_UR_TMP = 47, NV_UR_FLAG = 48, NV_CHUNK_IDX = 49, NV_DL = 50, NV_CDD_OUT = 51,
  NV_DECR_DEPTH = 52, NV_TMP_DOWN = 53, NV_M_IS_PARENT = 69, NV_TMP_IS_PAR = 85, NV_TMPIV = 101,
  NV_EQ_OUT = 109, NV_BIT_AT_DEPTH = 173, NV_NARROW_COUNT = 237,
  /* wide atoms */
  NV_ROOT_INV = 237, NV_E0_INV = 238, NV_E1_INV = 239, NV_ROOT_ISZ_IN = 240, NV_E0_ISZ_IN = 241,
  NV_E1_ISZ_IN = 242, NV_E1_IN1 = 243, NV_EQ_INV = 244, NV_EQ_ISZ_IN = 308, NV_EQ_IN1 = 372, NV_COUNT = 436
};
#define N_ATOMS (A_NV + NV_COUNT)

static const uint32_t IVW[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au,
                                0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
static const int SIGMA[16] = {2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8};
static const int GIDX[8][4] = {{0, 4, 8, 12}, {1, 5, 9, 13}, {2, 6, 10, 14}, {3, 7, 11, 15},
                               {0, 5, 10, 15}, {1, 6, 11, 12}, {2, 7, 8, 13}, {3, 4, 9, 14}};

/* ---------------------------------------------------------------- 256-bit field arithmetic */
static int fe_cmp(const fe *a, const fe *b) {
  for (int i = 3; i >= 0; i--) {
    if (a->l[i] < b->l[i]) return -1;
    if (a->l[i] > b->l[i]) return 1;
  }
  return 0;
}
static uint64_t fe_add_raw(fe *r, const fe *a, const fe *b) {
  u128 c = 0;
  for (int i = 0; i < 4; i++) { c += (u128)a->l[i] + b->l[i]; r->l[i] = (uint64_t)c; c >>= 64; }
  return (uint64_t)c;
}
static uint64_t fe_sub_raw(fe *r, const fe *a, const fe *b) {
  uint64_t br = 0;
  for (int i = 0; i < 4; i++) {
    u128 d = (u128)a->l[i] - b->l[i] - br;
    r->l[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1;
  }
  return br;
}
static fe fe_u64(uint64_t x) { fe r = {{x, 0, 0, 0}}; return r; }
static int fe_is_zero(const fe *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static fe fe_add(const field_t *F, fe a, fe b) {
  fe r; uint64_t c = fe_add_raw(&r, &a, &b);
  if (c || fe_cmp(&r, &F->p) >= 0) fe_sub_raw(&r, &r, &F->p);
  return r;
}
static fe fe_sub(const field_t *F, fe a, fe b) {
  fe r; if (fe_sub_raw(&r, &a, &b)) fe_add_raw(&r, &r, &F->p);
  return r;
}
/* Montgomery product a*b*R^-1 mod p (CIOS) */
static fe mont_mul(const field_t *F, const fe *a, const fe *b) {
  uint64_t t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    u128 c = 0;
    for (int j = 0; j < 4; j++) { c += (u128)a->l[j] * b->l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
    c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
    uint64_t m = t[0] * F->n0;
    c = (u128)m * F->p.l[0] + t[0]; c >>= 64;
    for (int j = 1; j < 4; j++) { c += (u128)m * F->p.l[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
    c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
  }
  fe r = {{t[0], t[1], t[2], t[3]}};
  if (t[4] || fe_cmp(&r, &F->p) >= 0) fe_sub_raw(&r, &r, &F->p);
  return r;
}
static fe fe_mul(const field_t *F, fe a, fe b) {
  fe t = mont_mul(F, &a, &b);          /* a*b/R */
  return mont_mul(F, &t, &F->r2);      /* *R */
}
/* in != 0 ? 1/in : 0   (circomlib IsZero) via Fermat */
static fe fe_inv(const field_t *F, fe a) {
  if (fe_is_zero(&a)) return a;
  fe e; fe two = fe_u64(2); fe_sub_raw(&e, &F->p, &two);
  fe am = mont_mul(F, &a, &F->r2);                 /* to Montgomery form */
  fe one = fe_u64(1);
  fe acc = mont_mul(F, &one, &F->r2);
  for (int i = 255; i >= 0; i--) {
    acc = mont_mul(F, &acc, &acc);
    if ((e.l[i >> 6] >> (i & 63)) & 1) acc = mont_mul(F, &acc, &am);
  }
  return mont_mul(F, &acc, &one);
}
static int fe_bit(const fe *a, int i) { return (int)((a->l[i >> 6] >> (i & 63)) & 1); }
/* value >> n == 0 ? */
static int fe_fits(const fe *a, int n) {
  const int limb = n >> 6, sh = n & 63;
  if (limb < 4 && sh && (a->l[limb] >> sh)) return 0;
  for (int i = limb + (sh ? 1 : 0); i < 4; i++) if (a->l[i]) return 0;
  return 1;
}
static void field_init(field_t *F, const uint64_t p[4]) {
  memcpy(F->p.l, p, 32);
  uint64_t inv = 1;
  for (int i = 0; i < 6; i++) inv *= 2 - p[0] * inv;      /* p^-1 mod 2^64 */
  F->n0 = (uint64_t)0 - inv;
  /* R^2 mod p by 512 modular doublings of 1 */
  fe x = fe_u64(1);
  for (int i = 0; i < 512; i++) x = fe_add(F, x, x);
  F->r2 = x;
}

static const uint64_t P_BN254[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
static const uint64_t P_VESTA[4] = {0x8c46eb2100000001ull, 0x224698fc0994a8ddull, 0x0ull, 0x4000000000000000ull};

/* ---------------------------------------------------------------- circuits */
typedef struct {
  int nwit;
  int ninputs;
  int is_nova;
  const uint64_t *prime;
  int32_t *slot_atom;   /* per slot */
  int16_t *slot_bit;    /* -1 whole */
  field_t F;
  int loaded;
} circuit_t;

static circuit_t CIRC[B3WO_NCIRCUITS] = {
  {24093, 28, 0, P_BN254, 0, 0, {{{0}}, {{0}}, 0}, 0},
  {23291, 32, 1, P_BN254, 0, 0, {{{0}}, {{0}}, 0}, 0},
  {23291, 32, 1, P_VESTA, 0, 0, {{{0}}, {{0}}, 0}, 0},
  {24614, 32, 1, P_BN254, 0, 0, {{{0}}, {{0}}, 0}, 0},
};

typedef struct { int failed; char msg[160]; } assert_t;
static void fail(assert_t *as, const char *what, int r, int g, int hf) {
  if (as->failed) return;
  as->failed = 1;
  snprintf(as->msg, sizeof as->msg, "Assert Failed. %s (round %d G %d half %d)", what, r, g, hf + 1);
}

static uint32_t rotr32(uint32_t x, int r) { return (x >> r) | (x << (32 - r)); }

/* Blake3Compression; inputs already in atoms[A_H..A_D]. Stops at the first failed assert. */
static void eval_compression(const field_t *F, fe *at, assert_t *as) {
  at[A_ONE] = fe_u64(1);
  fe v[16], msg[16], tmp[16];
  for (int i = 0; i < 8; i++) v[i] = at[A_H + i];
  for (int i = 0; i < 4; i++) v[8 + i] = fe_u64(IVW[i]);
  v[12] = at[A_T]; v[13] = at[A_T + 1]; v[14] = at[A_B]; v[15] = at[A_D];
  for (int i = 0; i < 16; i++) msg[i] = at[A_M + i];
  for (int r = 0; r < 7; r++) {
    for (int g = 0; g < 8; g++) {
      const int a = GIDX[g][0], b = GIDX[g][1], c = GIDX[g][2], d = GIDX[g][3];
      for (int hf = 0; hf < 2; hf++) {
        fe *o = &at[A_HG + 8 * ((r * 8 + g) * 2 + hf)];
        const int R1 = hf ? 8 : 16, R2 = hf ? 7 : 12;
        fe s1 = fe_add(F, fe_add(F, v[a], v[b]), msg[2 * g + hf]);     /* HalfFunG :89 */
        if (!fe_fits(&s1, 34)) { fail(as, "Bits34 inp === sum", r, g, hf); return; }   /* Bits34 :201 */
        uint32_t A = (uint32_t)s1.l[0];
        if (!fe_fits(&v[d], 32)) { fail(as, "ToBits(rxor2) inp === sum", r, g, hf); return; }  /* ToBits :153 */
        uint32_t DI = (uint32_t)v[d].l[0];
        uint32_t D2 = rotr32(DI ^ A, R1);                               /* RotXorBits :36-44 */
        fe s3 = fe_add(F, v[c], fe_u64(D2));                            /* HalfFunG :92 */
        if (!fe_fits(&s3, 33)) { fail(as, "Bits33 inp === sum", r, g, hf); return; }   /* Bits33 :176 */
        uint32_t C = (uint32_t)s3.l[0];
        if (!fe_fits(&v[b], 32)) { fail(as, "ToBits(rxor4) inp === sum", r, g, hf); return; }
        uint32_t BI = (uint32_t)v[b].l[0];
        uint32_t B4 = rotr32(BI ^ C, R2);
        o[HG_S1] = s1; o[HG_A] = fe_u64(A); o[HG_S3] = s3; o[HG_C] = fe_u64(C);
        o[HG_D2] = fe_u64(D2); o[HG_DI] = fe_u64(DI); o[HG_B4] = fe_u64(B4); o[HG_BI] = fe_u64(BI);
        v[a] = fe_u64(A); v[b] = fe_u64(B4); v[c] = fe_u64(C); v[d] = fe_u64(D2);
      }
    }
    for (int j = 0; j < 16; j++) tmp[j] = msg[SIGMA[j]];                /* Blake3Permute :24 */
    memcpy(msg, tmp, sizeof msg);
  }
  for (int k = 0; k < 16; k++) {                                        /* Blake3Compression :213-227 */
    fe x = v[k], y = k < 8 ? v[k + 8] : at[A_H + k - 8];
    if (!fe_fits(&x, 32)) { fail(as, "outXor ToBits x", 7, k, 0); return; }
    if (!fe_fits(&y, 32)) { fail(as, "outXor ToBits y", 7, k, 0); return; }
    at[A_O + k] = fe_u64((uint32_t)x.l[0] ^ (uint32_t)y.l[0]);
  }
}

/* IsZero(in): inv, out = 1 - in*inv */
static fe is_zero(const field_t *F, fe in, fe *inv) {
  *inv = fe_inv(F, in);
  return fe_sub(F, fe_u64(1), fe_mul(F, in, *inv));
}

static void eval_nova(const field_t *F, fe *at, assert_t *as) {
  fe *nv = at + A_NV;
  const fe one = fe_u64(1), zero = fe_u64(0);
  fe depth = nv[NV_DEPTH], leaf_depth = nv[NV_LEAF_DEPTH], total_depth = nv[NV_TOTAL_DEPTH];
  /* Blake3NovaTreePath_CheckDepth (:13-45; the Num2Bits(8) pair at :25-29 is absent from the committed WASMs) */
  nv[NV_ROOT_ISZ_IN] = fe_sub(F, zero, depth);
  nv[NV_IS_ROOT] = is_zero(F, nv[NV_ROOT_ISZ_IN], &nv[NV_ROOT_INV]);
  nv[NV_CP_IN1] = fe_sub(F, leaf_depth, one);
  nv[NV_CP_N2B_IN] = fe_sub(F, fe_add(F, depth, fe_u64(256)), nv[NV_CP_IN1]);     /* LessThan(8) */
  if (!fe_fits(&nv[NV_CP_N2B_IN], 9)) { fail(as, "check_parent Num2Bits(9)", -1, 0, 0); return; }
  fe is_parent = fe_u64(1 - fe_bit(&nv[NV_CP_N2B_IN], 8));
  nv[NV_IS_PARENT] = is_parent;
  nv[NV_ED_IN1] = fe_add(F, depth, one);
  nv[NV_ED_N2B_IN] = fe_sub(F, fe_add(F, leaf_depth, fe_u64(256)), nv[NV_ED_IN1]); /* GreaterEqThan(8) */
  if (!fe_fits(&nv[NV_ED_N2B_IN], 9)) { fail(as, "exceed_depth Num2Bits(9)", -1, 0, 0); return; }
  nv[NV_ED_OUT] = fe_u64(1 - fe_bit(&nv[NV_ED_N2B_IN], 8));
  if (!fe_is_zero(&nv[NV_ED_OUT])) { fail(as, "exceed_depth.out === 0 (CheckDepth line 38)", -1, 0, 0); return; }
  /* Blake3GetFlag (:122-167) */
  fe is_root = nv[NV_IS_ROOT];
  nv[NV_NOT_ROOT] = fe_sub(F, one, is_root);
  nv[NV_NOT_PARENT] = fe_sub(F, one, is_parent);
  nv[NV_E0_ISZ_IN] = fe_sub(F, zero, nv[NV_BLOCK_COUNT]);
  nv[NV_E0] = is_zero(F, nv[NV_E0_ISZ_IN], &nv[NV_E0_INV]);
  nv[NV_E1_IN1] = fe_sub(F, nv[NV_N_BLOCKS], one);
  nv[NV_E1_ISZ_IN] = fe_sub(F, nv[NV_E1_IN1], nv[NV_BLOCK_COUNT]);
  nv[NV_E1] = is_zero(F, nv[NV_E1_ISZ_IN], &nv[NV_E1_INV]);
  nv[NV_IS_LAST_BLOCK] = fe_mul(F, nv[NV_E1], nv[NV_NOT_PARENT]);
  nv[NV_FIRST] = fe_mul(F, nv[NV_E0], nv[NV_NOT_PARENT]);
  nv[NV_UR_TMP] = fe_sub(F, fe_add(F, is_parent, nv[NV_E1]), fe_mul(F, is_parent, nv[NV_E1]));   /* OR */
  nv[NV_UR_FLAG] = fe_mul(F, nv[NV_UR_TMP], is_root);
  fe dflag = nv[NV_FIRST];
  dflag = fe_add(F, dflag, fe_mul(F, fe_u64(2), nv[NV_IS_LAST_BLOCK]));
  dflag = fe_add(F, dflag, fe_mul(F, fe_u64(8), nv[NV_UR_FLAG]));
  dflag = fe_add(F, dflag, fe_mul(F, fe_u64(4), is_parent));
  /* Blake3GetDownLeftPath (:47-84) */
  fe chunk_idx = fe_add(F, nv[NV_CIL], fe_mul(F, nv[NV_CIH], fe_u64(1ull << 32)));
  nv[NV_CHUNK_IDX] = chunk_idx;
  if (!fe_fits(&chunk_idx, 65)) { fail(as, "down_left_path Num2Bits(65)", -1, 0, 0); return; }
  fe bad = zero;
  for (int i = 0; i < 64; i++) {
    nv[NV_EQ_IN1 + i] = fe_sub(F, total_depth, fe_u64((uint64_t)i + 2));
    nv[NV_EQ_ISZ_IN + i] = fe_sub(F, nv[NV_EQ_IN1 + i], depth);
    nv[NV_EQ_OUT + i] = is_zero(F, nv[NV_EQ_ISZ_IN + i], &nv[NV_EQ_INV + i]);
    fe nb = fe_u64(1 - fe_bit(&chunk_idx, i));
    bad = fe_add(F, bad, fe_mul(F, nb, nv[NV_EQ_OUT + i]));
    nv[NV_BIT_AT_DEPTH + i] = bad;
  }
  fe dl = fe_add(F, fe_sub(F, one, is_parent), fe_mul(F, is_parent, bad));
  nv[NV_DL] = dl;
  { fe chk = fe_mul(F, dl, fe_sub(F, one, dl));
    if (!fe_is_zero(&chk)) { fail(as, "down_left_path.out boolean", -1, 0, 0); return; } }
  /* Blake3GetFinal_m (:86-120) */
  fe ndl = fe_sub(F, one, dl), npar = nv[NV_NOT_PARENT];
  for (int i = 0; i < 16; i++) {
    fe td, mp;
    if (i < 8) {
      td = fe_mul(F, nv[NV_H + i], dl);
      mp = fe_add(F, fe_mul(F, nv[NV_M + i], ndl), td);
    } else {
      td = fe_mul(F, nv[NV_H + i - 8], ndl);
      mp = fe_add(F, fe_mul(F, nv[NV_M + i - 8], dl), td);
    }
    fe tp = fe_mul(F, mp, is_parent);
    nv[NV_TMP_DOWN + i] = td; nv[NV_M_IS_PARENT + i] = mp; nv[NV_TMP_IS_PAR + i] = tp;
    at[A_M + i] = fe_add(F, fe_mul(F, nv[NV_M + i], npar), tp);
  }
  /* Blake3Nova (:229-245) */
  for (int i = 0; i < 8; i++) {
    nv[NV_TMPIV + i] = fe_mul(F, fe_u64(IVW[i]), is_parent);
    at[A_H + i] = fe_add(F, fe_mul(F, nv[NV_H + i], npar), nv[NV_TMPIV + i]);
  }
  at[A_T] = fe_mul(F, nv[NV_CIL], npar);
  at[A_T + 1] = fe_mul(F, nv[NV_CIH], npar);
  at[A_B] = nv[NV_B];
  at[A_D] = dflag;
  eval_compression(F, at, as);
  if (as->failed) return;
  nv[NV_BLOCK_COUNT_OUT] = fe_add(F, nv[NV_BLOCK_COUNT], npar);                      /* :251 */
  nv[NV_CDD_OUT] = fe_sub(F, fe_add(F, nv[NV_IS_LAST_BLOCK], is_parent), fe_mul(F, nv[NV_IS_LAST_BLOCK], is_parent));
  nv[NV_DECR_DEPTH] = fe_mul(F, nv[NV_CDD_OUT], nv[NV_NOT_ROOT]);                    /* :258 */
  nv[NV_DEPTH_OUT] = fe_sub(F, depth, nv[NV_DECR_DEPTH]);                            /* :262 */
}

/* ---------------------------------------------------------------- public API */
int b3wo_load_layout(int circuit, const char *path) {
  if (circuit < 0 || circuit >= B3WO_NCIRCUITS) return -1;
  circuit_t *c = &CIRC[circuit];
  FILE *f = fopen(path, "r");
  if (!f) return -2;
  int32_t *sa = (int32_t *)malloc(sizeof(int32_t) * c->nwit);
  int16_t *sb = (int16_t *)malloc(sizeof(int16_t) * c->nwit);
  for (int i = 0; i < c->nwit; i++) sa[i] = -1;
  char line[256];
  while (fgets(line, sizeof line, f)) {
    int s, a, b0, len;
    if (line[0] == 'W' && sscanf(line + 1, "%d %d %d", &s, &a, &len) == 3) {
      for (int j = 0; j < len; j++) { if (s + j >= c->nwit) goto bad; sa[s + j] = a + j; sb[s + j] = -1; }
    } else if (line[0] == 'B' && sscanf(line + 1, "%d %d %d %d", &s, &a, &b0, &len) == 4) {
      for (int j = 0; j < len; j++) { if (s + j >= c->nwit) goto bad; sa[s + j] = a; sb[s + j] = (int16_t)(b0 + j); }
    }
  }
  fclose(f);
  for (int i = 0; i < c->nwit; i++) if (sa[i] < 0 || sa[i] >= N_ATOMS) goto bad2;
  free(c->slot_atom); free(c->slot_bit);
  c->slot_atom = sa; c->slot_bit = sb;
  field_init(&c->F, c->prime);
  c->loaded = 1;
  return 0;
bad:
  fclose(f);
bad2:
  free(sa); free(sb);
  return -3;
}

int b3wo_witness_size(int circuit) { return CIRC[circuit].nwit; }
int b3wo_input_size(int circuit) { return CIRC[circuit].ninputs; }
void b3wo_prime(int circuit, uint8_t out[32]) { memcpy(out, CIRC[circuit].prime, 32); }

/* inputs: ninputs field elements, 32 B little-endian each, already reduced mod p, in record order
 *   compression: h[8] m[16] t[2] b d           nova: n_blocks block_count h[8] chunk_idx_low
 *   chunk_idx_high leaf_depth total_depth depth m[16] b
 * body: nwit*32 bytes.  Returns 0, or 4 (circom "Assert Failed") with a description in err. */
int b3wo_witness(int circuit, const uint8_t *inputs_le32, uint8_t *body, char *err, size_t errlen) {
  circuit_t *c = &CIRC[circuit];
  if (!c->loaded) { if (err && errlen) snprintf(err, errlen, "layout not loaded"); return -1; }
  fe *at = (fe *)calloc(N_ATOMS, sizeof(fe));
  assert_t as; as.failed = 0; as.msg[0] = 0;
  fe in[32];
  for (int i = 0; i < c->ninputs; i++) memcpy(in[i].l, inputs_le32 + 32 * i, 32);
  if (!c->is_nova) {
    for (int i = 0; i < 28; i++) at[A_H + i] = in[i];     /* H,M,T,B,D are contiguous atoms 1..28 */
    eval_compression(&c->F, at, &as);
  } else {
    for (int i = 0; i < 32; i++) at[A_NV + i] = in[i];    /* nova inputs are NV+0..31 in record order */
    eval_nova(&c->F, at, &as);
  }
  int rc = 0;
  if (as.failed) {
    if (err && errlen) snprintf(err, errlen, "%s", as.msg);
    rc = 4;
  } else {
    for (int s = 0; s < c->nwit; s++) {
      const fe *a = &at[c->slot_atom[s]];
      if (c->slot_bit[s] < 0) memcpy(body + 32 * (size_t)s, a->l, 32);
      else { memset(body + 32 * (size_t)s, 0, 32); body[32 * (size_t)s] = (uint8_t)fe_bit(a, c->slot_bit[s]); }
    }
  }
  free(at);
  return rc;
}

/* canonical u32 records (the batch fast-path input format) */
int b3wo_witness_u32(int circuit, const uint32_t *rec, uint8_t *body, char *err, size_t errlen) {
  uint8_t in[32 * 32];
  memset(in, 0, sizeof in);
  for (int i = 0; i < CIRC[circuit].ninputs; i++) memcpy(in + 32 * i, &rec[i], 4);
  return b3wo_witness(circuit, in, body, err, errlen);
}

/* n records -> n bodies, single thread; returns number of failed asserts */
int b3wo_witness_batch_u32(int circuit, const uint32_t *recs, int n, uint8_t *bodies) {
  int bad = 0;
  const circuit_t *c = &CIRC[circuit];
  for (int i = 0; i < n; i++)
    if (b3wo_witness_u32(circuit, recs + (size_t)i * c->ninputs, bodies + (size_t)i * c->nwit * 32, 0, 0)) bad++;
  return bad;
}

/* .wtns v2 header, 76 bytes (witness_calculator.js:208-262) */
void b3wo_wtns_header(int circuit, uint8_t out[76]) {
  const circuit_t *c = &CIRC[circuit];
  uint32_t w[19];
  memcpy(&w[0], "wtns", 4);
  w[1] = 2; w[2] = 2; w[3] = 1; w[4] = 40; w[5] = 0; w[6] = 32;
  memcpy(&w[7], c->prime, 32);
  w[15] = (uint32_t)c->nwit; w[16] = 2;
  uint64_t len = 32ull * (uint64_t)c->nwit;
  w[17] = (uint32_t)len; w[18] = (uint32_t)(len >> 32);
  memcpy(out, w, 76);
}
