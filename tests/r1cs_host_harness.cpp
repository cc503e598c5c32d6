// r1cs_host_harness.cpp — test harness of hot-proofs-blake3-circom_amd/csrc/b3w_r1cs_host.cpp, compiled by
// tests/test_r1cs_host_asan_cpu.py with AddressSanitizer + UBSan (no GPU, no HIP):
//   harness <image> <nwit> <mutations> <seed>
// 1. the pristine image must load; every index the kernels would follow is checked to lie inside its array, and the lean term
//    stream (parts sorted, bit runs folded) must say the same as the gather formulation: for pseudo-random small z, every part
//    of every row sums to the same value (mod 2^64, coefficients taken by their low 64 bits) in both.
// 2. `mutations` mutated copies (truncations, flipped bytes, patched counts and wire numbers) are either refused with a text
//    or load as a different system that passes the same index checks.  The sanitizers watch the whole time.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <stdexcept>
#include <vector>

#include "b3w_r1cs_host.h"

#define CHECK(c)                                                                 \
  do {                                                                           \
    if (!(c)) { fprintf(stderr, "invariant failed at line %d: %s\n", __LINE__, #c); return false; } \
  } while (0)

static uint64_t rng_state;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

static bool indices_ok(const B3wR1csHost &H) {
  const uint32_t T = B3W_R1CS_TILE;
  CHECK(H.rowdesc.size() == 4 * (size_t)H.m && H.row_id.size() == H.m && H.cids.size() == H.wires.size());
  CHECK(H.coefR.size() == 16 * (size_t)H.ncoef);
  for (uint32_t k = 0; k < H.m; k++) {
    const uint64_t end = (uint64_t)H.rowdesc[4 * k] + H.rowdesc[4 * k + 1] + H.rowdesc[4 * k + 2] + H.rowdesc[4 * k + 3];
    CHECK(end <= H.wires.size() && H.row_id[k] < H.m);
  }
  for (size_t i = 0; i < H.wires.size(); i++) CHECK(H.wires[i] < H.nwires && H.cids[i] < H.ncoef);
  if (!H.tiled) return true;
  CHECK(H.ntiles == (H.nwires + T - 1) / T && H.tdesc.size() == 4 * (size_t)H.ntiles && H.ttdesc.size() == 2 * (size_t)H.ntiles);
  CHECK(H.ltdesc.size() == 2 * (size_t)H.ntiles && H.coef_small.size() == H.ncoef && H.max_ext <= T);
  CHECK(H.trows.size() == H.lrows.size() && H.trow_id.size() * 4 == H.trows.size() && H.trow_k.size() == H.trow_id.size());
  CHECK(H.lterms.size() % 4 == 0);
  for (uint32_t t = 0; t < H.ntiles; t++) {
    const uint32_t row0 = H.tdesc[4 * t], nrows = H.tdesc[4 * t + 1], ext0 = H.tdesc[4 * t + 2], next = H.tdesc[4 * t + 3];
    CHECK((uint64_t)row0 + nrows <= H.trow_id.size() && (uint64_t)ext0 + next <= H.text.size() && next <= H.max_ext && nrows <= H.max_tile_rows);
    for (uint32_t j = 0; j < next; j++) CHECK(H.text[ext0 + j] < H.nwires);
    const uint32_t lt0 = H.ltdesc[2 * t], ltn = H.ltdesc[2 * t + 1], tt0 = H.ttdesc[2 * t], ttn = H.ttdesc[2 * t + 1];
    CHECK(lt0 % 4 == 0 && (uint64_t)lt0 + ltn + 2 <= H.lterms.size() && ltn <= H.max_lean_terms);       // (two read-ahead words behind)
    CHECK((uint64_t)tt0 + ttn + 1 <= H.tterms.size() && ttn <= H.max_tile_terms);
    for (uint32_t r = row0; r < row0 + nrows; r++) {
      CHECK(H.trow_k[r] < H.m && H.trow_id[r] < H.m);
      for (int lean = 0; lean < 2; lean++) {
        const std::vector<uint32_t> &rows = lean ? H.lrows : H.trows;
        const std::vector<uint32_t> &terms = lean ? H.lterms : H.tterms;
        const uint32_t base = lean ? lt0 : tt0, cnt = lean ? ltn : ttn;
        const uint32_t off = rows[4 * r], ya = rows[4 * r + 1], nb = rows[4 * r + 2], w3 = rows[4 * r + 3];
        const bool boolean = ya >> 31;
        const uint32_t na = ya & (lean ? 0x3FFFFFFFu : 0x7FFFFFFFu), nc = boolean ? 0u : w3;      // (lean rows: bit 30 = "always deferred")
        if (lean && (ya & 0x40000000u)) CHECK(!boolean);
        if (boolean) CHECK(w3 < T + next);
        CHECK(off >= base && (uint64_t)off + na + nb + (boolean && !lean ? 0u : nc) <= (uint64_t)base + cnt);
        const uint32_t total = na + nb + nc;
        for (uint32_t q = 0; q < total; q++) {
          const uint32_t w = terms[off + q];
          if (lean && (w >> 16) == 0xFFFFu) {
            CHECK(q + 1 < total);
            const uint32_t w1 = terms[off + q + 1], n = w1 & 0xFFu, sh = (w1 >> 8) & 0xFFu;
            CHECK(n >= 4 && n <= 64 && sh + n <= 62 && (w & 0xFFFFu) + n <= T + next);
            // a run lies inside one part
            const uint32_t part_end = q < na ? na : q < na + nb ? na + nb : total;
            CHECK(q + 1 < part_end);
            q++;
          } else {
            CHECK((w & 0xFFFFu) < T + next && (w >> 16) < H.ncoef);
          }
        }
      }
    }
  }
  // the stream kernel's program: every index it follows inside its array
  CHECK(H.srows.size() == H.lrows.size() && H.sgdesc.size() == 4 * (size_t)H.ntiles && H.sgwords.size() == H.sgmeta.size());
  CHECK(H.smask_groups == (T + ((H.max_ext + 32u) & ~31u) + 63u) / 64u && H.smask.size() == (size_t)H.ntiles * H.smask_groups);
  CHECK(H.scost.size() == (size_t)H.ntiles + 1 && H.scost[0] == 0);
  for (uint32_t t = 0; t < H.ntiles; t++) CHECK(H.scost[t + 1] > H.scost[t] && H.scost[t + 1] - H.scost[t] < (1ull << 24));
  for (uint32_t t = 0; t < H.ntiles; t++) {
    const uint32_t row0 = H.tdesc[4 * t], nrows = H.tdesc[4 * t + 1], next = H.tdesc[4 * t + 3];
    const uint32_t gw0 = H.sgdesc[4 * t], gwn = H.sgdesc[4 * t + 1], ng = H.sgdesc[4 * t + 2];
    const uint32_t gruns = H.sgdesc[4 * t + 3];            // entries that are bit runs: they stand first, padded to whole chunks of 64
    CHECK((uint64_t)gw0 + gwn + 1 <= H.sgwords.size() && gwn <= H.max_g_words && ng <= H.max_g_rows && gruns <= gwn && gruns % 64 == 0);
    uint32_t seen_g = 0, last_class = 0;
    for (uint32_t r = row0; r < row0 + nrows; r++) {
      const uint32_t x = H.srows[4 * r], y = H.srows[4 * r + 1], z = H.srows[4 * r + 2], w = H.srows[4 * r + 3];
      // the rows of a tile stand in the order general | truth table | always deferred | booleanity (the kernel's waves 0, 1 own the general rows)
      const uint32_t cls = (y >> 28) == 1u ? 0u : (y >> 29) == 1u ? 1u : (y >> 31) ? 3u : 2u;
      CHECK(cls >= last_class);
      last_class = cls;
      auto masked = [&](uint32_t idx) { return (H.smask[(size_t)t * H.smask_groups + (idx >> 6)] >> (idx & 63u)) & 1ull; };
      if (y >> 31) { CHECK(w < T + next && (H.lrows[4 * r + 1] >> 31) && H.lrows[4 * r + 3] == w && masked(w)); continue; }
      CHECK(!(H.lrows[4 * r + 1] >> 31));
      if ((y >> 29) == 1u) {
        const uint32_t k = (y >> 16) & 7u;
        CHECK(k >= 1 && k <= 5);
        const uint32_t idx[5] = {x & 0xFFFFu, x >> 16, z & 0xFFFFu, z >> 16, y & 0xFFFFu};
        for (uint32_t j = 0; j < 5; j++) CHECK(j < k ? idx[j] < T + next && masked(idx[j]) : idx[j] == 0);
      } else if ((y >> 28) == 1u) {
        CHECK(x == seen_g && x < ng);
        seen_g++;
      } else CHECK(y == 0x40000000u);
    }
    CHECK(seen_g == ng);
    for (uint32_t i = 0; i < gwn; i++) {
      const uint32_t w = H.sgwords[gw0 + i], mt = H.sgmeta[gw0 + i];
      if (mt & 4u) { CHECK(i < gruns && mt == 4u); continue; }                      // a null entry: only in the runs' padding
      CHECK((mt & 3u) < 3u && (mt >> 8) < ng && ((mt >> 3) & 1u) == (i < gruns ? 1u : 0u) && !(mt & 0xF0u));
      if (i < gruns) {                                     // first element | length << 16 | shift << 23 | negative << 29
        const uint32_t n = (w >> 16) & 0x7Fu, sh = (w >> 23) & 0x3Fu;
        CHECK(n >= 4 && n <= 64 && (w & 0xFFFFu) + n <= T + next && sh + n <= 62 && !(w >> 30));
      } else CHECK((w & 0xFFFFu) < T + next && (w >> 16) < H.ncoef);
    }
  }
  return true;
}

// the walk kernel's program: every index it follows inside its array, every import from an EARLIER tile
static bool walk_ok(const B3wR1csHost &H) {
  if (!H.walk) return true;
  const uint32_t T = B3W_R1CS_TILE, nt = H.wunits;          // units: a tile each, a tile with many general rows several
  CHECK(nt >= H.ntiles && nt <= B3W_WALK_MAX_UNITS);
  CHECK(H.wtile.size() == (size_t)B3W_WT_WORDS * nt && H.wmask.size() == 16 * (size_t)nt && H.wtiles4.size() == 4 * (size_t)nt);
  CHECK(H.wexp_slots % 64 == 0 && H.wexp_slots <= B3W_WALK_MAX_EXP_SLOTS && H.wstatic.size() == (size_t)nt * H.wstatic_words);
  CHECK(H.wrow_k.size() == H.m && H.wrow_id.size() == H.m && H.wruns.size() % 4 == 0 && H.wrun_row.size() + 64 == H.wruns.size() / 4);
  CHECK(H.went_w.size() == H.went_m.size() && H.wmax_gen <= B3W_WALK_MAX_GEN && H.wmax_ent <= B3W_WALK_MAX_ENT);
  std::vector<uint8_t> seen(H.m, 0);
  uint32_t rows_so_far = 0, slots_so_far = 0, prev_src = 0xFFFFFFFFu;
  for (uint32_t t = 0; t < nt; t++) {
    const uint32_t *w = &H.wtile[(size_t)B3W_WT_WORDS * t];
    const uint32_t n_local = w[B3W_WT_NLOCAL], exp_n = w[B3W_WT_EXP_N], slot0 = w[B3W_WT_EXP_SLOT0], src = w[B3W_WT_SRC];
    const bool first = src != prev_src;                     // the tile's first unit: exports, runs, booleanity rows
    CHECK(src < H.ntiles && (first ? src == prev_src + 1u : true) && (t != 0 || src == 0));
    prev_src = src;
    CHECK(n_local == std::min<uint32_t>(T, H.nwires - src * T) && slot0 % 64 == 0 && (first ? slot0 == slots_so_far : (exp_n == 0 && w[B3W_WT_RUN_N] == 0)));
    if (!first) {                                           // the same mask as the tile's first unit, and nothing but general rows
      for (int q = 0; q < 16; q++) CHECK(H.wmask[16 * (size_t)t + q] == H.wmask[16 * (size_t)(t - 1) + q]);
      CHECK(w[B3W_WT_GEN_N] == w[B3W_WT_NROWS]);
    }
    CHECK(w[B3W_WT_GEN_N] <= B3W_WALK_SPLIT_GEN);
    CHECK((uint64_t)w[B3W_WT_EXP_OFF] + exp_n + 64 <= H.wexp.size() && exp_n <= H.wmax_exp);
    for (uint32_t j = 0; j < exp_n; j++) CHECK(H.wexp[w[B3W_WT_EXP_OFF] + j] < n_local && (j == 0 || H.wexp[w[B3W_WT_EXP_OFF] + j] > H.wexp[w[B3W_WT_EXP_OFF] + j - 1]));
    slots_so_far += (exp_n + 63u) & ~63u;
    // an element index of this tile: local, or a slot an EARLIER tile has filled
    auto idx_ok = [&](uint32_t idx, uint32_t len) { return idx < T ? idx + len <= n_local : idx - T + len <= slot0; };      // (slot0: the TILE's first slot)
    CHECK(w[B3W_WT_ROW0] == rows_so_far && H.wtiles4[4 * t] == rows_so_far && H.wtiles4[4 * t + 1] == w[B3W_WT_NROWS]);
    const uint32_t row0 = w[B3W_WT_ROW0], nrows = w[B3W_WT_NROWS], gen_n = w[B3W_WT_GEN_N];
    CHECK(nrows <= H.wmax_rows && (nrows + 63) / 64 <= H.wstatic_words && gen_n <= nrows && gen_n <= H.wmax_gen);
    for (uint32_t r = row0; r < row0 + nrows; r++) { CHECK(H.wrow_k[r] < H.m && !seen[H.wrow_k[r]]); seen[H.wrow_k[r]] = 1; }
    rows_so_far += nrows;
    const uint32_t run_off = w[B3W_WT_RUN_OFF], run_n = w[B3W_WT_RUN_N];
    CHECK((uint64_t)run_off + run_n + 64 <= H.wruns.size() / 4 && run_n <= H.wmax_runs);
    uint32_t next_row = row0 + gen_n;
    for (uint32_t r = run_off; r < run_off + run_n; r++) {
      const uint32_t *d = &H.wruns[4 * (size_t)r];
      const uint32_t len = ((d[3] >> 16) & 31u) + 1u, nops = (d[3] >> 21) & 7u, strides = (d[3] >> 24) & 31u;
      const uint32_t idx[5] = {d[1] & 0xFFFFu, d[1] >> 16, d[2] & 0xFFFFu, d[2] >> 16, d[3] & 0xFFFFu};
      CHECK(nops >= 1 && nops <= 5 && !(d[3] >> 29) && (len > 1 || strides == 0));
      for (uint32_t o = 0; o < 5; o++) CHECK(o < nops ? idx_ok(idx[o], (strides >> o) & 1u ? len : 1u) : (idx[o] == 0 && !((strides >> o) & 1u)));
      CHECK(H.wrun_row[r] == next_row);                     // the rows of the runs follow the general rows, run after run
      next_row += len;
    }
    CHECK(next_row <= row0 + nrows);
    const uint32_t ent_off = w[B3W_WT_ENT_OFF], ent_n = w[B3W_WT_ENT_N], ent_runs = w[B3W_WT_ENT_RUNS];
    CHECK((uint64_t)ent_off + ent_n + 64 <= H.went_w.size() && ent_n <= H.wmax_ent && ent_runs <= ent_n && ent_runs % 64 == 0);
    for (uint32_t i = 0; i < ent_n; i++) {
      const uint32_t e = H.went_w[ent_off + i], mt = H.went_m[ent_off + i];
      if (mt & 4u) { CHECK(i < ent_runs && mt == 4u); continue; }
      CHECK((mt & 3u) < 3u && (mt >> 8) < gen_n && ((mt >> 3) & 1u) == (i < ent_runs ? 1u : 0u) && !(mt & 0xF0u));
      if (i < ent_runs) {
        const uint32_t n = (e >> 16) & 0x7Fu, sh = (e >> 23) & 0x3Fu;
        CHECK(n >= 4 && n <= 64 && sh + n <= 62 && !(e >> 30) && idx_ok(e & 0xFFFFu, n));
      } else if (e >> 31) {                                // a term with a coefficient +-2^k: element | k << 16 | negative << 22 | 1 << 31
        CHECK(idx_ok(e & 0xFFFFu, 1) && ((e >> 16) & 63u) <= 62u && !((e >> 23) & 0xFFu));
      } else CHECK(idx_ok(e & 0xFFFFu, 1) && (e >> 16) < H.ncoef);
    }
    // always-deferred rows lie behind the general rows and the runs, inside the tile
    for (uint32_t k = 0; k < H.wstatic_words; k++) {
      const unsigned long long sm = H.wstatic[(size_t)t * H.wstatic_words + k];
      for (uint32_t bit = 0; bit < 64; bit++) if ((sm >> bit) & 1ull) CHECK(64 * k + bit >= next_row - row0 && 64 * k + bit < nrows);
    }
  }
  CHECK(rows_so_far == H.m && slots_so_far == H.wexp_slots && prev_src + 1u == H.ntiles);
  // the always-deferred rows as the deferred kernel's list: every row's unique pairs, expanded by their part flags, are the row's terms
  {
    std::vector<uint32_t> srows;
    for (uint32_t u = 0; u < nt; u++)
      for (uint32_t pos = 0; pos < H.wtiles4[4 * u + 1]; pos++)
        if ((H.wstatic[(size_t)u * H.wstatic_words + (pos >> 6)] >> (pos & 63u)) & 1ull) srows.push_back(H.wtiles4[4 * u] + pos);
    const std::vector<uint32_t> &L = H.wstatic_list;
    CHECK(H.wstatic_ids.size() == srows.size() && L.size() >= 4 * srows.size() && L.size() % 2 == 0);
    size_t next_pair = 2 * srows.size();
    for (size_t i = 0; i < srows.size(); i++) {
      const uint32_t k = H.wrow_k[srows[i]];
      CHECK(H.wstatic_ids[i] == H.wrow_id[srows[i]]);
      const uint32_t first = L[4 * i], n = L[4 * i + 1];
      CHECK(first == next_pair && (size_t)first + n <= L.size() / 2);
      next_pair += n;
      const uint32_t *d = &H.rowdesc[4 * (size_t)k];
      CHECK(L[4 * i + 2] == (d[1] == 0 || d[2] == 0 ? 1u : 0u) && L[4 * i + 3] == (d[3] ? 1u : 0u) && n <= d[1] + d[2] + d[3]);
      uint32_t q = d[0];
      for (uint32_t part = 0; part < 3; part++) {
        std::vector<uint64_t> want, got;
        for (uint32_t x = 0; x < d[1 + part]; x++, q++) want.push_back((uint64_t)H.wires[q] << 16 | H.cids[q]);
        for (uint32_t j = first; j < first + n; j++) {
          const uint32_t w = L[2 * (size_t)j], mt = L[2 * (size_t)j + 1];
          CHECK(w < H.nwires && (mt & 0xFFFFu) < H.ncoef && !(mt >> 19) && (mt >> 16) != 0);
          if ((mt >> (16 + part)) & 1u) got.push_back((uint64_t)w << 16 | (mt & 0xFFFFu));
        }
        std::sort(want.begin(), want.end()); std::sort(got.begin(), got.end());
        CHECK(want == got);
      }
    }
    CHECK(next_pair == L.size() / 2);
  }
  return true;
}

// the lean stream against the gather arrays
static bool same_sums(const B3wR1csHost &H, const uint8_t prime_le[32]) {
  if (!H.tiled) return true;
  const uint32_t T = B3W_R1CS_TILE;
  uint64_t p_lo;
  memcpy(&p_lo, prime_le, 8);
  std::vector<uint64_t> z(H.nwires);
  for (auto &v : z) { const uint64_t x = rnd(); v = (x & 3) ? (x >> 2) & 1 : (x >> 8) & 0xFFFFFFFFu; }
  auto coef_lo = [&](uint32_t cid) { uint64_t c; memcpy(&c, &H.coefR[16 * (size_t)cid], 8); return c; };
  for (uint32_t t = 0; t < H.ntiles; t++) {
    const uint32_t row0 = H.tdesc[4 * t], nrows = H.tdesc[4 * t + 1], ext0 = H.tdesc[4 * t + 2];
    auto wire_of = [&](uint32_t idx) { return idx < T ? t * T + idx : H.text[ext0 + idx - T]; };
    for (uint32_t r = row0; r < row0 + nrows; r++) {
      const uint32_t k = H.trow_k[r];
      const uint32_t g_off = H.rowdesc[4 * k], g_n[3] = {H.rowdesc[4 * k + 1], H.rowdesc[4 * k + 2], H.rowdesc[4 * k + 3]};
      const uint32_t off = H.lrows[4 * r], ya = H.lrows[4 * r + 1];
      const bool boolean = ya >> 31;
      const uint32_t l_n[3] = {ya & 0x3FFFFFFFu, H.lrows[4 * r + 2], boolean ? 0u : H.lrows[4 * r + 3]};
      if (!boolean) {                                        // the "always deferred" flag = the row has a coefficient that is no small integer
        bool not_small = false;
        for (uint32_t x = 0; x < g_n[0] + g_n[1] + g_n[2]; x++) not_small = not_small || H.coef_small[H.cids[g_off + x]] == B3W_R1CS_NOT_SMALL;
        CHECK(not_small == ((ya & 0x40000000u) != 0));
      }
      if (boolean) CHECK(g_n[0] == 1 && g_n[1] == 2 && g_n[2] == 0 && wire_of(H.lrows[4 * r + 3]) == H.wires[g_off]);
      uint32_t gq = g_off, lq = off;
      for (int part = 0; part < 3; part++) {
        uint64_t want = 0, got = 0;
        for (uint32_t x = 0; x < g_n[part]; x++, gq++) want += coef_lo(H.cids[gq]) * z[H.wires[gq]];
        for (uint32_t x = 0; x < l_n[part]; x++, lq++) {
          const uint32_t w = H.lterms[lq];
          if ((w >> 16) == 0xFFFFu) {
            const uint32_t w1 = H.lterms[lq + 1], n = w1 & 0xFFu, sh = (w1 >> 8) & 0xFFu;
            const bool neg = (w1 >> 16) & 1u;
            for (uint32_t i = 0; i < n; i++) {
              const uint64_t c = neg ? p_lo - (1ull << (sh + i)) : 1ull << (sh + i);
              got += c * z[wire_of((w & 0xFFFFu) + i)];
            }
            x++; lq++;
          } else {
            got += coef_lo(w >> 16) * z[wire_of(w & 0xFFFFu)];
          }
        }
        CHECK(want == got);
      }
    }
    // the stream program's general rows: the words of row g, part by part, against the gather arrays
    const uint32_t gw0 = H.sgdesc[4 * t], gwn = H.sgdesc[4 * t + 1], ng = H.sgdesc[4 * t + 2];
    std::vector<uint64_t> sums(3 * (size_t)ng, 0);
    for (uint32_t i = 0; i < gwn; i++) {
      const uint32_t w = H.sgwords[gw0 + i], mt = H.sgmeta[gw0 + i];
      if (mt & 4u) continue;
      uint64_t v = 0;
      if (mt & 8u) {
        const uint32_t n = (w >> 16) & 0x7Fu, sh = (w >> 23) & 0x3Fu;
        for (uint32_t q = 0; q < n; q++) v += ((w >> 29) & 1u ? p_lo - (1ull << (sh + q)) : 1ull << (sh + q)) * z[wire_of((w & 0xFFFFu) + q)];
      } else v = coef_lo(w >> 16) * z[wire_of(w & 0xFFFFu)];
      sums[3 * (size_t)(mt >> 8) + (mt & 3u)] += v;
    }
    for (uint32_t r = row0; r < row0 + nrows; r++) {
      if ((H.srows[4 * r + 1] >> 28) != 1u) continue;
      const uint32_t g = H.srows[4 * r], k = H.trow_k[r];
      uint32_t gq = H.rowdesc[4 * k];
      for (int part = 0; part < 3; part++) {
        uint64_t want = 0;
        for (uint32_t x = 0; x < H.rowdesc[4 * k + 1 + part]; x++, gq++) want += coef_lo(H.cids[gq]) * z[H.wires[gq]];
        CHECK(want == sums[3 * (size_t)g + part]);
      }
    }
  }
  return true;
}

// the walk program's data path on a random assignment: exports, general rows' sums (mod 2^64, against the gather arrays), every row
// of every truth-table run against the row evaluated in plain integers, booleanity rows' wires in the must-be-bit masks
static bool walk_same(const B3wR1csHost &H, const uint8_t prime_le[32]) {
  if (!H.walk) return true;
  const uint32_t T = B3W_R1CS_TILE, nt = H.wunits;
  uint64_t p_lo;
  memcpy(&p_lo, prime_le, 8);
  auto coef_lo = [&](uint32_t cid) { uint64_t c; memcpy(&c, &H.coefR[16 * (size_t)cid], 8); return c; };
  std::vector<uint32_t> unit_of(H.ntiles, 0);               // a tile's first unit
  for (uint32_t u = nt; u-- > 0;) unit_of[H.wtile[(size_t)B3W_WT_WORDS * u + B3W_WT_SRC]] = u;
  std::vector<uint64_t> z(H.nwires);
  for (uint32_t w = 0; w < H.nwires; w++) {
    const uint64_t x = rnd();
    const bool mustbit = (H.wmask[(size_t)unit_of[w / T] * 16 + ((w % T) >> 6)] >> (w & 63u)) & 1ull;
    z[w] = mustbit || (x & 3) ? (x >> 2) & 1 : (x >> 8) & 0xFFFFFFFFu;
  }
  z[0] = 1;
  std::vector<uint64_t> xel(H.wexp_slots, 0);
  std::vector<uint32_t> slot_wire(H.wexp_slots, 0xFFFFFFFFu);
  size_t tt_rows = 0;
  for (uint32_t u = 0; u < nt; u++) {
    const uint32_t *w = &H.wtile[(size_t)B3W_WT_WORDS * u];
    const uint32_t t = w[B3W_WT_SRC];                       // (the tile; u indexes what is per unit)
    auto value = [&](uint32_t idx) { return idx < T ? z[t * T + idx] : xel[idx - T]; };
    auto wire_of = [&](uint32_t idx) { return idx < T ? t * T + idx : slot_wire[idx - T]; };
    const uint32_t row0 = w[B3W_WT_ROW0], nrows = w[B3W_WT_NROWS], gen_n = w[B3W_WT_GEN_N];
    // general rows
    std::vector<uint64_t> sums(3 * (size_t)gen_n, 0);
    for (uint32_t i = 0; i < w[B3W_WT_ENT_N]; i++) {
      const uint32_t e = H.went_w[w[B3W_WT_ENT_OFF] + i], mt = H.went_m[w[B3W_WT_ENT_OFF] + i];
      if (mt & 4u) continue;
      uint64_t v = 0;
      if (mt & 8u) {
        const uint32_t n = (e >> 16) & 0x7Fu, sh = (e >> 23) & 0x3Fu;
        for (uint32_t q = 0; q < n; q++) {
          CHECK(value((e & 0xFFFFu) + q) <= 1);             // a run's elements are must-be-bit wires
          v += ((e >> 29) & 1u ? p_lo - (1ull << (sh + q)) : 1ull << (sh + q)) * value((e & 0xFFFFu) + q);
        }
      } else if (e >> 31) v = ((e >> 22) & 1u ? p_lo - (1ull << ((e >> 16) & 63u)) : 1ull << ((e >> 16) & 63u)) * value(e & 0xFFFFu);
      else v = coef_lo(e >> 16) * value(e & 0xFFFFu);
      sums[3 * (size_t)(mt >> 8) + (mt & 3u)] += v;
    }
    for (uint32_t g = 0; g < gen_n; g++) {
      const uint32_t k = H.wrow_k[row0 + g];
      uint32_t gq = H.rowdesc[4 * k];
      for (int part = 0; part < 3; part++) {
        uint64_t want = 0;
        for (uint32_t x = 0; x < H.rowdesc[4 * k + 1 + part]; x++, gq++) { want += coef_lo(H.cids[gq]) * z[H.wires[gq]]; CHECK(H.wires[gq] / T <= t); }
        CHECK(want == sums[3 * (size_t)g + part]);
      }
    }
    // truth-table runs
    uint32_t next_row = row0 + gen_n;
    for (uint32_t r = w[B3W_WT_RUN_OFF]; r < w[B3W_WT_RUN_OFF] + w[B3W_WT_RUN_N]; r++) {
      const uint32_t *d = &H.wruns[4 * (size_t)r];
      const uint32_t len = ((d[3] >> 16) & 31u) + 1u, nops = (d[3] >> 21) & 7u, strides = (d[3] >> 24) & 31u;
      const uint32_t idx[5] = {d[1] & 0xFFFFu, d[1] >> 16, d[2] & 0xFFFFu, d[2] >> 16, d[3] & 0xFFFFu};
      for (uint32_t j = 0; j < len; j++, next_row++) {
        uint32_t a = 0;
        std::vector<uint32_t> ops;
        for (uint32_t o = 0; o < nops; o++) {
          const uint32_t ix = idx[o] + ((strides >> o) & 1u ? j : 0u);
          CHECK(value(ix) <= 1);
          a |= (uint32_t)value(ix) << o;
          ops.push_back(wire_of(ix));
        }
        const bool holds = (d[0] >> a) & 1u;
        const uint32_t k = H.wrow_k[next_row];
        uint32_t gq = H.rowdesc[4 * k];
        __int128 part[3] = {0, 0, 0};
        bool small = true;
        for (int pt = 0; pt < 3; pt++)
          for (uint32_t x = 0; x < H.rowdesc[4 * k + 1 + pt]; x++, gq++) {
            const long long cs = H.coef_small[H.cids[gq]];
            small = small && cs != B3W_R1CS_NOT_SMALL;
            part[pt] += (__int128)cs * (__int128)z[H.wires[gq]];
            CHECK(std::find(ops.begin(), ops.end(), H.wires[gq]) != ops.end());      // the run's operands are this row's wires
          }
        if (small) CHECK(holds == (part[0] * part[1] == part[2]));
        tt_rows++;
      }
    }
    // behind them: always-deferred rows (a coefficient that is no small integer, or very long), then booleanity rows
    for (uint32_t r = next_row; r < row0 + nrows; r++) {
      const uint32_t k = H.wrow_k[r], pos = r - row0;
      const bool stat = (H.wstatic[(size_t)u * H.wstatic_words + (pos >> 6)] >> (pos & 63u)) & 1ull;
      if (stat) continue;
      CHECK(H.rowdesc[4 * k + 1] == 1 && H.rowdesc[4 * k + 2] == 2 && H.rowdesc[4 * k + 3] == 0);
      const uint32_t bw = H.wires[H.rowdesc[4 * k]];
      CHECK(bw / T == t && ((H.wmask[(size_t)u * 16 + ((bw % T) >> 6)] >> (bw & 63u)) & 1ull));
    }
    // this tile's exports, for the tiles behind it
    for (uint32_t j = 0; j < w[B3W_WT_EXP_N]; j++) {
      xel[w[B3W_WT_EXP_SLOT0] + j] = z[t * T + H.wexp[w[B3W_WT_EXP_OFF] + j]];
      slot_wire[w[B3W_WT_EXP_SLOT0] + j] = t * T + H.wexp[w[B3W_WT_EXP_OFF] + j];
    }
  }
  printf("walk program: %zu truth-table rows in runs agree with plain integers\n", tt_rows);
  return true;
}

int main(int argc, char **argv) {
  if (argc < 5) { fprintf(stderr, "usage: harness <image> <nwit> <mutations> <seed>\n"); return 2; }
  FILE *f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 2; }
  std::vector<uint8_t> img;
  uint8_t buf[65536];
  size_t n;
  while ((n = fread(buf, 1, sizeof buf, f)) > 0) img.insert(img.end(), buf, buf + n);
  fclose(f);
  const uint32_t nwit = (uint32_t)strtoul(argv[2], nullptr, 10);
  const int mutations = atoi(argv[3]);
  rng_state = strtoull(argv[4], nullptr, 10) * 0x9E3779B97F4A7C15ull + 1;
  if (img.size() < 64) { fprintf(stderr, "image too short\n"); return 2; }
  uint8_t prime[32];
  memcpy(prime, &img[12 + 12 + 4], 32);                    // "r1cs" ver nsec | type size(8) | fieldSize | prime   (header section first)
  {
    B3wR1csHost H;
    if (!b3w_r1cs_host_build(img.data(), img.size(), prime, nwit, &H)) { fprintf(stderr, "pristine image refused: %s\n", H.error.c_str()); return 1; }
    if (!indices_ok(H) || !walk_ok(H) || !same_sums(H, prime) || !walk_same(H, prime)) return 1;
    printf("walk %d: %u units over %u tiles, %u export slots, %zu runs, %zu entries, %u general rows a unit at most\n", (int)H.walk, H.wunits, H.ntiles, H.wexp_slots, H.wruns.size() / 4, H.went_w.size(), H.wmax_gen);
    uint64_t runs = 0;
    for (size_t i = 0; i + 1 < H.lterms.size(); i++) runs += (H.lterms[i] >> 16) == 0xFFFFu;
    printf("pristine: %u constraints, %llu terms, tiled %d, %u tiles, max_ext %u, lean words %zu (%llu runs)\n", H.m,
           (unsigned long long)H.nterms, (int)H.tiled, H.ntiles, H.max_ext, H.lterms.size(), (unsigned long long)runs);
  }
  int refused = 0, loaded = 0;
  for (int k = 0; k < mutations; k++) {
    std::vector<uint8_t> bad(img);
    switch (k % 5) {
      case 0: bad.resize(rnd() % bad.size()); break;
      case 1: for (int i = 0, e = 1 + (int)(rnd() % 3); i < e; i++) bad[rnd() % 100] = (uint8_t)rnd(); break;           // header bytes
      case 2: { const size_t pos = 100 + rnd() % (bad.size() - 104); const uint32_t v = (uint32_t)rnd(); memcpy(&bad[pos], &v, 4); break; }
      case 3: { const uint32_t v = (rnd() & 1) ? 0xFFFFFFFFu : (uint32_t)(rnd() % 100000); memcpy(&bad[24 + 36 + 24], &v, 4); break; }     // mConstraints
      default: for (int i = 0; i < 8; i++) bad[100 + rnd() % (bad.size() - 100)] ^= (uint8_t)(1u << (rnd() & 7)); break;  // bit flips in the constraints
    }
    try {
      B3wR1csHost H;
      if (b3w_r1cs_host_build(bad.data(), bad.size(), prime, nwit, &H)) {
        if (!indices_ok(H) || !walk_ok(H)) { fprintf(stderr, "mutation %d loaded with bad indices\n", k); return 1; }
        loaded++;
      } else {
        if (H.error.empty()) { fprintf(stderr, "mutation %d refused without a text\n", k); return 1; }
        refused++;
      }
    } catch (const std::bad_alloc &) { refused++; } catch (const std::length_error &) { refused++; }
  }
  printf("mutations: %d refused, %d loaded\n", refused, loaded);
  return 0;
}
