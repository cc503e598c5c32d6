// b3w_r1cs_defs.h — what the constraint check's host code (b3w_r1cs_host.cpp, no HIP) and its kernels share
#pragma once
#include <stdint.h>

struct B3wField {
  uint32_t p[8];      // modulus of the circuit's field, little-endian limbs
  uint32_t r2[8];     // 2^512 mod p
  uint32_t inv;       // -p^-1 mod 2^32
};
#define B3W_R1CS_TILE 1024u
#define B3W_R1CS_NOT_SMALL ((long long)0x8000000000000000ull)

// the walk program's per-tile table (B3wR1csHost::wtile, 16 words per tile)
#define B3W_WT_NLOCAL 0     // elements of the tile (1 024 but for the last)
#define B3W_WT_EXP_OFF 1    // first entry of its export list
#define B3W_WT_EXP_N 2      // exports
#define B3W_WT_EXP_SLOT0 3  // first slot of the export area it writes (a multiple of 64)
#define B3W_WT_RUN_OFF 4    // first truth-table run
#define B3W_WT_RUN_N 5
#define B3W_WT_ENT_OFF 6    // first entry of the general rows
#define B3W_WT_ENT_N 7      // entries (bit runs first, padded to whole chunks of 64, then terms)
#define B3W_WT_ENT_RUNS 8   // entries that are bit runs (a multiple of 64)
#define B3W_WT_GEN_N 9      // general rows: the tile's first rows
#define B3W_WT_ROW0 10      // first row of the tile in the walk row order
#define B3W_WT_NROWS 11
#define B3W_WT_SRC 12        // the tile whose elements the unit reads (a tile with many general rows is several units)
#define B3W_WT_WORDS 16
#define B3W_WALK_MAX_EXP_SLOTS 4096u
#define B3W_WALK_SPLIT_GEN 320u   // general rows per unit at most: a tile with more becomes several units (b3w_r1cs_host.cpp)
#define B3W_WALK_MAX_UNITS 56u    // units per body at most (the body word of the deferred kernel: one bit a unit, 8 bits for the wide records)
#define B3W_WALK_MAX_GEN 512u
#define B3W_WALK_MAX_ENT 4096u
#define B3W_WALK_WIDE_CAP 224u     // wide records per body (an O2 nova step has 66, the circomkit build 197; the body word counts them in 8 bits)
