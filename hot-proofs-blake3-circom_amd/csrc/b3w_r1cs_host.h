// b3w_r1cs_host.h — what b3w_r1cs_host_build makes of an iden3 .r1cs image (see b3w_r1cs_host.cpp); layouts as documented at
// the launch declarations in b3w_kernels.h.  No HIP in here.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "b3w_r1cs_defs.h"

struct B3wR1csHost {
  std::string error;                                       // why the image was refused
  uint32_t m = 0, nwires = 0, npubout = 0, npubin = 0, nprvin = 0;
  uint64_t nterms = 0;
  B3wField field{};
  // gather formulation
  std::vector<uint32_t> rowdesc, row_id, wires, coefR;
  std::vector<uint16_t> cids;
  // tile formulations (tiled: every tile of B3W_R1CS_TILE wires needs at most that many outside wires)
  bool tiled = false;
  uint32_t ntiles = 0, max_ext = 0, max_tile_terms = 0, max_tile_rows = 0, max_lean_terms = 0, ncoef = 0;
  std::vector<long long> coef_small;
  std::vector<uint32_t> tdesc, ttdesc, text, trows, trow_id, trow_k, tterms;      // 32-byte tile kernel
  std::vector<uint32_t> ltdesc, lrows, lterms;                                    // lean kernel: its own rows and term stream (bit runs folded)
  // stream kernel: one descriptor per tile row (same order as lrows; classes booleanity / truth table / always deferred / general),
  // per tile {first general entry, entries, general rows, run entries (padded to whole chunks of 64)}; the general rows' ENTRIES,
  // one lane each — sgwords: a term = element | coefficient id << 16, a bit run = first element | length << 16 | shift << 23 |
  // negative << 29; sgmeta: part (0 A, 1 B, 2 C) | 4 = null padding entry | 8 = bit run | general row << 8
  std::vector<uint32_t> srows, sgdesc, sgwords, sgmeta;
  uint32_t max_g_words = 0, max_g_rows = 0;
  std::vector<unsigned long long> smask;          // per tile x smask_groups: LDS elements the tile's booleanity / truth-table rows take for bits
  uint32_t smask_groups = 0;
  std::vector<unsigned long long> scost;          // ntiles + 1 prefix sums of the tiles' relative unit costs (stream kernel's work split)
  // ---- the WALK program (b3w_r1cs_walk_kernel, round 4): a workgroup walks whole bodies, tile after tile.  A row belongs to the
  // tile of its HIGHEST wire, so everything else it mentions lies in the same or an EARLIER tile: such an earlier wire is
  // "exported" by its home tile into a per-body export area in LDS when that tile is in LDS anyway — no outside wire is ever
  // gathered from HBM.  Element index of a row: < TILE = local, TILE + slot = export slot (each tile's slots start at a multiple of
  // 64 so that the same number indexes the bit-packed copy).
  bool walk = false;                               // the system fits the walk kernel
  uint32_t wunits = 0;                             // UNITS per body: a tile each, but a tile with more than B3W_WALK_SPLIT_GEN general rows is several
                                                   // (B3W_WT_SRC = the tile a unit reads); every "per tile" below is per unit
  uint32_t wlinear_rows = 0;                       // rows without A or B terms: an optimiser leaves none; an unsimplified system keeps its differences as wires, and
                                                   // small NEGATIVE values (p - k) with them: the walk kernel then runs its SIGNED instantiation
  uint32_t wexp_slots = 0;                         // export area, in slots (padded)
  uint32_t wmax_gen = 0, wmax_ent = 0, wmax_exp = 0, wmax_runs = 0, wmax_rows = 0, wstatic_words = 0;
  std::vector<uint32_t> wtile;                     // 16 per tile: see B3W_WT_* in b3w_r1cs_defs.h
  std::vector<unsigned long long> wmask;           // 16 per tile: local elements some row takes for a bit
  std::vector<uint16_t> wexp;                      // export lists: local element numbers, in slot order
  std::vector<uint32_t> wruns;                     // truth-table RUNS, 4 words each: table | idx0, idx1 | idx2, idx3 | idx4, len - 1 << 16, k << 21, strides << 24
  std::vector<uint32_t> wrun_row;                  // per run: its first row in the walk row order (rows of a run are consecutive)
  std::vector<uint32_t> went_w, went_m;            // general rows' entries (term or bit run) and meta words, as sgwords / sgmeta
  std::vector<uint32_t> wrow_k, wrow_id;           // walk row order -> gather row, -> constraint number in the file
  std::vector<uint32_t> wtiles4;                   // per tile {first row, rows, 0, 0}: the deferred kernel's view of the walk order
  std::vector<unsigned long long> wstatic;         // per tile wstatic_words words: rows that are ALWAYS deferred (bit = row - first)
  // the same rows as a list the deferred kernel walks for every body: per row four words {first pair, pairs, linear, has C terms},
  // behind them the rows' UNIQUE terms as pairs {wire, coefficient id | parts << 16} (bit 16 A, 17 B, 18 C) — a term that stands in
  // several parts with the same coefficient (the always-deferred row of a nova step is X (X - 1) = 0 with an X of 66 field-sized
  // terms) is multiplied once; and the rows' constraint numbers
  std::vector<uint32_t> wstatic_list, wstatic_ids;
};

// false: refused, H->error says why.  May throw std::bad_alloc / std::length_error on absurd sizes (the caller catches).
bool b3w_r1cs_host_build(const uint8_t *img, size_t len, const uint8_t prime_le[32], uint32_t nwit, B3wR1csHost *H);
