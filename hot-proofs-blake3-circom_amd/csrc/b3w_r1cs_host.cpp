// b3w_r1cs_host.cpp — host side of the constraint check: an iden3 .r1cs image (UNTRUSTED input) -> the arrays the kernels of
// b3w_r1cs.hip take.  Plain C++, no HIP: b3w_r1cs_api.cpp uploads the result; tests/test_r1cs_host_asan_cpu.py compiles this file
// alone with AddressSanitizer + UBSan and runs mutated images through it.
#include "b3w_r1cs_host.h"

#include <algorithm>
#include <array>
#include <map>
#include <string.h>

namespace {
// little-endian reader over the file image; `ok` goes false on the first read past the end
struct ByteReader {
  const uint8_t *p; size_t len, pos = 0; bool ok = true;
  ByteReader(const uint8_t *p_, size_t n) : p(p_), len(n) {}
  bool need(size_t n) { if (!ok || len - pos < n) ok = false; return ok; }
  uint32_t u32() { if (!need(4)) return 0; uint32_t v; memcpy(&v, p + pos, 4); pos += 4; return v; }
  uint64_t u64() { if (!need(8)) return 0; uint64_t v; memcpy(&v, p + pos, 8); pos += 8; return v; }
  const uint8_t *bytes(size_t n) { if (!need(n)) return nullptr; const uint8_t *q = p + pos; pos += n; return q; }
};

void u256_add_mod(uint32_t a[8], const uint32_t b[8], const uint32_t p[8]) {
  uint64_t c = 0;
  for (int i = 0; i < 8; i++) { const uint64_t t = (uint64_t)a[i] + b[i] + c; a[i] = (uint32_t)t; c = t >> 32; }
  bool ge = c != 0;
  if (!ge) { ge = true; for (int i = 7; i >= 0; --i) if (a[i] != p[i]) { ge = a[i] > p[i]; break; } }
  if (ge) { uint64_t br = 0; for (int i = 0; i < 8; i++) { const uint64_t t = (uint64_t)a[i] - p[i] - br; a[i] = (uint32_t)t; br = (t >> 63) & 1; } }
}
// a * b / 2^256 mod p (CIOS; a, b < p; inv = -p^-1 mod 2^32) — exact field products for the truth tables of the stream program
void mont_mul_host(uint32_t out[8], const uint32_t a[8], const uint32_t b[8], const uint32_t p[8], uint32_t inv) {
  uint32_t t[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 8; i++) {
    uint64_t c = 0;
    for (int j = 0; j < 8; j++) { const uint64_t s = (uint64_t)a[j] * b[i] + t[j] + c; t[j] = (uint32_t)s; c = s >> 32; }
    uint64_t s = (uint64_t)t[8] + c;
    t[8] = (uint32_t)s; t[9] = (uint32_t)(s >> 32);
    const uint32_t m = t[0] * inv;
    c = ((uint64_t)m * p[0] + t[0]) >> 32;
    for (int j = 1; j < 8; j++) { const uint64_t s2 = (uint64_t)m * p[j] + t[j] + c; t[j - 1] = (uint32_t)s2; c = s2 >> 32; }
    s = (uint64_t)t[8] + c;
    t[7] = (uint32_t)s;
    t[8] = t[9] + (uint32_t)(s >> 32);
  }
  bool ge = t[8] != 0;
  if (!ge) { ge = true; for (int i = 7; i >= 0; --i) if (t[i] != p[i]) { ge = t[i] > p[i]; break; } }
  if (ge) { uint64_t br = 0; for (int i = 0; i < 8; i++) { const uint64_t d = (uint64_t)t[i] - p[i] - br; t[i] = (uint32_t)d; br = (d >> 63) & 1; } }
  memcpy(out, t, 32);
}
// x * 2^256 mod p by 256 modular doublings (host, set-up only)
void to_montgomery_host(uint32_t x[8], const uint32_t p[8]) {
  for (int i = 0; i < 256; i++) { uint32_t y[8]; memcpy(y, x, 32); u256_add_mod(x, y, p); }
}
}  // namespace

bool b3w_r1cs_host_build(const uint8_t *img, size_t len, const uint8_t prime_le[32], uint32_t nwit, B3wR1csHost *H) {
  auto bad = [&](const char *why) { H->error = std::string("r1cs: ") + why; return false; };
  // iden3 r1cs binary format v1: "r1cs" | u32 version | u32 nSections | { u32 type | u64 size | body }*
  ByteReader rd(img, len);
  const uint8_t *magic = rd.bytes(4);
  if (!magic || memcmp(magic, "r1cs", 4)) return bad("not an r1cs file");
  if (rd.u32() != 1) return bad("unsupported format version");
  const uint32_t nsec = rd.u32();
  size_t hdr_at = 0, cons_at = 0, cons_len = 0;
  for (uint32_t s = 0; s < nsec && rd.ok; s++) {
    const uint32_t type = rd.u32();
    const uint64_t size = rd.u64();
    if (!rd.ok || size > len - rd.pos) return bad("truncated section");
    if (type == 1) hdr_at = rd.pos;
    if (type == 2) { cons_at = rd.pos; cons_len = (size_t)size; }
    if (type == 4 || type == 5) return bad("custom gates are not supported");
    rd.pos += (size_t)size;
  }
  if (!hdr_at || !cons_at) return bad("header or constraint section missing");
  ByteReader h(img + hdr_at, len - hdr_at);
  if (h.u32() != 32) return bad("field size must be 32 bytes");
  const uint8_t *prime = h.bytes(32);
  if (!prime || memcmp(prime, prime_le, 32)) return bad("the file's prime is not this circuit's field");
  const uint32_t nwires = h.u32(), npubout = h.u32(), npubin = h.u32(), nprvin = h.u32();
  (void)h.u64();                                          // nLabels
  const uint32_t m = h.u32();
  if (!h.ok) return bad("truncated header");
  if (nwires != nwit) return bad("nWires differs from this circuit's witness size");
  if ((uint64_t)m * 12 > cons_len) return bad("more constraints announced than the constraint section can hold");
  // constraints: A, B, C as (u32 wire, 32-byte LE coefficient) lists; distinct coefficients are tabulated
  uint32_t P[8];
  memcpy(P, prime_le, 32);
  uint32_t pm1[8];
  memcpy(pm1, P, 32);
  pm1[0] -= 1;                                            // p is odd: no borrow
  std::vector<std::array<uint32_t, 8>> coefs(2);          // ids 0 (+1) and 1 (-1) are handled without a multiplication
  coefs[0] = {1, 0, 0, 0, 0, 0, 0, 0};
  memcpy(coefs[1].data(), pm1, 32);
  std::map<std::array<uint32_t, 8>, uint32_t> coef_id;
  coef_id[coefs[0]] = 0;
  coef_id[coefs[1]] = 1;
  struct Row { uint32_t off, na, nb, nc, id; };
  std::vector<Row> rows;
  rows.reserve(m);
  std::vector<uint32_t> wires;
  std::vector<uint16_t> cids;
  ByteReader c(img + cons_at, cons_len);
  for (uint32_t k = 0; k < m; k++) {
    Row r{(uint32_t)wires.size(), 0, 0, 0, k};
    uint32_t *cnt[3] = {&r.na, &r.nb, &r.nc};
    for (int part = 0; part < 3; part++) {
      const uint32_t n = c.u32();
      if (!c.ok || (uint64_t)n * 36 > cons_len - c.pos) return bad("truncated constraint section");
      for (uint32_t t = 0; t < n; t++) {
        const uint32_t w = c.u32();
        std::array<uint32_t, 8> cf;
        memcpy(cf.data(), c.bytes(32), 32);
        if (w >= nwires) return bad("wire index out of range");
        bool ge = true;
        for (int i = 7; i >= 0; --i) if (cf[i] != P[i]) { ge = cf[i] > P[i]; break; }
        if (ge) return bad("coefficient not reduced mod p");
        bool zero = true;
        for (int i = 0; i < 8; i++) zero &= cf[i] == 0;
        if (zero) continue;
        auto it = coef_id.find(cf);
        uint32_t id;
        if (it == coef_id.end()) {
          id = (uint32_t)coefs.size();
          if (id > 0xFFFF) return bad("more than 65 536 distinct coefficients");
          coef_id[cf] = id;
          coefs.push_back(cf);
        } else id = it->second;
        wires.push_back(w);
        cids.push_back((uint16_t)id);
        (*cnt[part])++;
      }
    }
    if (wires.size() > 0xFFFFFFF0ull) return bad("too many terms");
    rows.push_back(r);
  }
  // rows of one shape side by side: the lanes of a wave then run the same trip counts
  std::stable_sort(rows.begin(), rows.end(), [](const Row &a, const Row &b) {
    if (a.nc != b.nc) return a.nc > b.nc;
    if (a.na != b.na) return a.na > b.na;
    return a.nb > b.nb;
  });
  std::vector<uint32_t> rowdesc(4 * (size_t)m), row_id(m);
  for (uint32_t k = 0; k < m; k++) {
    rowdesc[4 * k] = rows[k].off; rowdesc[4 * k + 1] = rows[k].na; rowdesc[4 * k + 2] = rows[k].nb; rowdesc[4 * k + 3] = rows[k].nc;
    row_id[k] = rows[k].id;
  }
  // ---- tile formulation: a row belongs to the tile most of its wires lie in (the constant wire 0 does not vote); the
  // wires it mentions outside that tile are the tile's "outside wires", staged into LDS behind the tile
  const uint32_t T = B3W_R1CS_TILE, ntiles = (nwires + T - 1) / T;
  std::vector<std::vector<uint32_t>> tile_rows(ntiles);
  std::vector<std::map<uint32_t, uint32_t>> tile_ext(ntiles);     // outside wire -> its number in the tile
  for (uint32_t t = 1; t < ntiles; t++) tile_ext[t][0] = 0;       // the constant wire is outside wire 0 of every tile but the first
  {
    std::vector<uint32_t> votes(ntiles);
    for (uint32_t k = 0; k < m; k++) {                            // rows[] is shape-sorted: tile_rows keeps that order
      const Row &r = rows[k];
      const uint32_t nt = r.na + r.nb + r.nc;
      std::fill(votes.begin(), votes.end(), 0u);
      uint32_t best = 0;
      for (uint32_t t = 0; t < nt; t++) { const uint32_t w = wires[r.off + t]; if (w) votes[w / T]++; }
      for (uint32_t t = 1; t < ntiles; t++) if (votes[t] > votes[best]) best = t;
      tile_rows[best].push_back(k);
      for (uint32_t t = 0; t < nt; t++) {
        const uint32_t w = wires[r.off + t];
        if (w / T != best && !tile_ext[best].count(w)) { const uint32_t id = (uint32_t)tile_ext[best].size(); tile_ext[best][w] = id; }
      }
    }
  }
  // a tile's outside wires in ADDRESS order (the map's order; the constant wire stays number 0): the kernels gather them a lane per
  // wire, and neighbours in the list that are neighbours in the body — the bits of a word another tile owns — share cache lines
  for (uint32_t t = 0; t < ntiles; t++) {
    uint32_t id = 0;
    for (auto &kv : tile_ext[t]) kv.second = id++;
  }
  uint32_t max_ext = 0;
  for (uint32_t t = 0; t < ntiles; t++) max_ext = std::max<uint32_t>(max_ext, (uint32_t)tile_ext[t].size());
  uint32_t longest = 0;
  for (const Row &r : rows) longest = std::max(longest, r.na + r.nb + r.nc);
  // (the tile kernel sums small terms as 128-bit integers: rows stay below 2^20 terms)
  const bool tiled = max_ext <= T && coefs.size() <= 0xFFFF && longest < (1u << 20);
  // coefficients as small signed integers (c or c - p), for the tile kernel's integer path
  std::vector<long long> coef_small(coefs.size(), B3W_R1CS_NOT_SMALL);
  for (size_t i = 0; i < coefs.size(); i++) {
    const std::array<uint32_t, 8> &cf = coefs[i];
    uint32_t hi = 0;
    for (int q = 2; q < 8; q++) hi |= cf[q];
    const uint64_t lo64 = (uint64_t)cf[0] | (uint64_t)cf[1] << 32;
    if (!hi && lo64 < (1ull << 62)) { coef_small[i] = (long long)lo64; continue; }
    uint32_t neg[8];                                                               // p - c
    uint64_t br = 0;
    for (int q = 0; q < 8; q++) { const uint64_t d = (uint64_t)P[q] - cf[q] - br; neg[q] = (uint32_t)d; br = (d >> 63) & 1; }
    hi = 0;
    for (int q = 2; q < 8; q++) hi |= neg[q];
    const uint64_t n64 = (uint64_t)neg[0] | (uint64_t)neg[1] << 32;
    if (!hi && n64 < (1ull << 62)) coef_small[i] = -(long long)n64;
  }
  std::vector<uint32_t> tdesc(4 * (size_t)ntiles), ttdesc(2 * (size_t)ntiles), text, trows, trow_id, trow_k, tterms;
  uint32_t max_tile_terms = 0, max_tile_rows = 0;
  // the lean kernel's own term stream: the same rows, each part sorted by LDS index, with BIT RUNS folded — four or more terms
  // over consecutive elements whose coefficients are +-2^k, +-2^(k+1), ... (the recomposition rows "word = sum 2^i bit_i" of a
  // circom circuit, 26 % of the terms of blake3_compression and 43 % of the O2 nova systems) become two words:
  // idx0 | 0xFFFF << 16, then n | k << 8 | negative << 16.  The kernel evaluates a run from the tile's bit-packed elements.
  std::vector<uint32_t> ltdesc(2 * (size_t)ntiles), lrows, lterms;
  uint32_t max_lean_terms = 0;
  std::vector<uint8_t> in_run(nwires, 0);                  // wires that stand in a bit run: bits of a recomposed word
  std::vector<uint32_t> cur_ext;                           // outside wires of the tile emit_part is working on
  uint32_t cur_tile = 0;
  auto pow2 = [&](uint16_t cid, bool &neg, uint32_t &k) {
    const long long c = coef_small[cid];
    if (c == B3W_R1CS_NOT_SMALL || c == 0) return false;
    const unsigned long long mag = c < 0 ? 0ull - (unsigned long long)c : (unsigned long long)c;
    if (mag & (mag - 1)) return false;
    neg = c < 0;
    k = (uint32_t)__builtin_ctzll(mag);
    return true;
  };
  auto emit_part = [&](std::vector<std::pair<uint32_t, uint16_t>> &part) -> uint32_t {       // returns the words emitted
    std::stable_sort(part.begin(), part.end(), [](const std::pair<uint32_t, uint16_t> &a, const std::pair<uint32_t, uint16_t> &b) { return a.first < b.first; });
    const size_t before = lterms.size();
    for (size_t i = 0; i < part.size();) {
      bool neg = false; uint32_t k0 = 0;
      size_t j = i + 1;
      if (pow2(part[i].second, neg, k0)) {
        while (j < part.size() && j - i < 64 && part[j].first == part[j - 1].first + 1) {
          bool ng = false; uint32_t kk = 0;
          if (!pow2(part[j].second, ng, kk) || ng != neg || kk != k0 + (uint32_t)(j - i)) break;
          j++;
        }
      }
      if (j - i >= 4) {
        for (size_t q = i; q < j; q++) {
          const uint32_t idx = part[q].first;
          in_run[idx < T ? cur_tile * T + idx : cur_ext[idx - T]] = 1;
        }
        lterms.push_back(part[i].first | 0xFFFF0000u);
        lterms.push_back((uint32_t)(j - i) | k0 << 8 | (neg ? 1u << 16 : 0u));
        i = j;
      } else {
        lterms.push_back(part[i].first | (uint32_t)part[i].second << 16);
        i++;
      }
    }
    return (uint32_t)(lterms.size() - before);
  };
  auto build_tiles = [&]() {
    text.clear(); trows.clear(); trow_id.clear(); trow_k.clear(); tterms.clear(); lrows.clear(); lterms.clear();
    max_tile_terms = max_tile_rows = max_lean_terms = 0;
    for (uint32_t t = 0; t < ntiles; t++) {
      while (lterms.size() & 3) lterms.push_back(0);      // a tile's list starts on 16 bytes (the lean kernel stages it in uint4s)
      ltdesc[2 * t] = (uint32_t)lterms.size();
      ttdesc[2 * t] = (uint32_t)tterms.size();
      tdesc[4 * t] = (uint32_t)(trows.size() / 4); tdesc[4 * t + 1] = (uint32_t)tile_rows[t].size();
      tdesc[4 * t + 2] = (uint32_t)text.size(); tdesc[4 * t + 3] = (uint32_t)tile_ext[t].size();
      std::vector<uint32_t> ext(tile_ext[t].size());
      for (const auto &kv : tile_ext[t]) ext[kv.second] = kv.first;
      text.insert(text.end(), ext.begin(), ext.end());
      cur_ext = ext; cur_tile = t;
      for (uint32_t k : tile_rows[t]) {
        const Row &r = rows[k];
        // booleanity:  A = {w: 1},  B = {wire 0: 1, w: -1} or {w: 1, wire 0: -1},  C = {}
        bool boolean = r.na == 1 && r.nb == 2 && r.nc == 0 && cids[r.off] == 0 && wires[r.off] != 0;
        if (boolean) {
          const uint32_t w = wires[r.off], w1 = wires[r.off + 1], w2 = wires[r.off + 2];
          const uint16_t c1 = cids[r.off + 1], c2 = cids[r.off + 2];
          boolean = (w1 == 0 && w2 == w && ((c1 == 0 && c2 == 1) || (c1 == 1 && c2 == 0))) ||
                    (w2 == 0 && w1 == w && ((c2 == 0 && c1 == 1) || (c2 == 1 && c1 == 0)));
        }
        uint32_t bool_idx = 0;                                 // a booleanity row names its element by LDS index
        if (boolean) { const uint32_t w_a = wires[r.off]; bool_idx = w_a / T == t ? w_a - t * T : T + tile_ext[t].at(w_a); }
        trows.push_back((uint32_t)tterms.size()); trows.push_back(r.na | (boolean ? 0x80000000u : 0u)); trows.push_back(r.nb);
        trows.push_back(boolean ? bool_idx : r.nc);
        trow_id.push_back(r.id);
        trow_k.push_back(k);
        const uint32_t lean_off = (uint32_t)lterms.size();
        uint32_t lean_n[3] = {0, 0, 0};
        const uint32_t part_len[3] = {r.na, r.nb, r.nc};
        uint32_t q = 0;
        for (int part = 0; part < 3; part++) {
          std::vector<std::pair<uint32_t, uint16_t>> terms_of_part;
          for (uint32_t x = 0; x < part_len[part]; x++, q++) {
            const uint32_t w = wires[r.off + q];
            const uint32_t idx = w / T == t ? w - t * T : T + tile_ext[t][w];
            tterms.push_back(idx | (uint32_t)cids[r.off + q] << 16);
            terms_of_part.emplace_back(idx, cids[r.off + q]);
          }
          lean_n[part] = emit_part(terms_of_part);
        }
        // a row with a coefficient that is no small integer cannot be decided by the integer kernels: flagged (bit 30), so that they
        // hand it to the deferred kernel without walking its terms (the O2 nova systems have one such row of 66 ... 133 terms)
        bool not_small = false;
        for (uint32_t x = 0; x < r.na + r.nb + r.nc; x++) not_small = not_small || coef_small[cids[r.off + x]] == B3W_R1CS_NOT_SMALL;
        lrows.push_back(lean_off); lrows.push_back(lean_n[0] | (boolean ? 0x80000000u : not_small ? 0x40000000u : 0u)); lrows.push_back(lean_n[1]);
        lrows.push_back(boolean ? bool_idx : lean_n[2]);
      }
      ttdesc[2 * t + 1] = (uint32_t)tterms.size() - ttdesc[2 * t];
      max_tile_terms = std::max(max_tile_terms, ttdesc[2 * t + 1]);
      ltdesc[2 * t + 1] = (uint32_t)lterms.size() - ltdesc[2 * t];
      max_lean_terms = std::max(max_lean_terms, ltdesc[2 * t + 1]);
      max_tile_rows = std::max<uint32_t>(max_tile_rows, (uint32_t)tile_rows[t].size());
    }
    tterms.push_back(0);
    lterms.push_back(0); lterms.push_back(0);             // the lean kernel fetches up to two term words ahead,
    while (lterms.size() & 3) lterms.push_back(0);        // and stages whole uint4s
  };
  // A row's class in the stream program (below): 0 general, 1 truth table, 2 always deferred, 3 booleanity.  `nw` = its lean words.
  auto boolean_wire = [&](const Row &r, uint32_t *w_out) {    // A = {w: 1},  B = {wire 0: 1, w: -1} or {w: 1, wire 0: -1},  C = {}
    if (!(r.na == 1 && r.nb == 2 && r.nc == 0 && cids[r.off] == 0 && wires[r.off] != 0)) return false;
    const uint32_t w = wires[r.off], w1 = wires[r.off + 1], w2 = wires[r.off + 2];
    const uint16_t c1 = cids[r.off + 1], c2 = cids[r.off + 2];
    const bool ok = (w1 == 0 && w2 == w && ((c1 == 0 && c2 == 1) || (c1 == 1 && c2 == 0))) ||
                    (w2 == 0 && w1 == w && ((c2 == 0 && c1 == 1) || (c2 == 1 && c1 == 0)));
    if (ok) *w_out = w;
    return ok;
  };
  std::vector<uint8_t> is_bit;                             // wires expected to hold bits (set once the bit runs are known)
  auto row_class = [&](const Row &r, uint32_t nw, std::vector<uint32_t> *W_out) -> int {
    uint32_t bw = 0;
    if (boolean_wire(r, &bw)) return 3;
    const uint32_t nt = r.na + r.nb + r.nc;
    std::vector<uint32_t> W;                               // truth table: at most five distinct wires, all of them bits by their own constraints
    bool all_bits = true;
    for (uint32_t x = 0; x < nt && all_bits && W.size() <= 5; x++) {
      const uint32_t w = wires[r.off + x];
      all_bits = is_bit[w] != 0;
      if (std::find(W.begin(), W.end(), w) == W.end()) W.push_back(w);
    }
    if (all_bits && W.size() <= 5 && nt <= 64) { if (W_out) *W_out = W; return 1; }
    bool not_small = false;
    for (uint32_t x = 0; x < nt; x++) not_small = not_small || coef_small[cids[r.off + x]] == B3W_R1CS_NOT_SMALL;
    return not_small || nw > 256 ? 2 : 0;
  };
  if (tiled) {
    // Built TWICE.  The first pass finds the bit runs and every row's lean words, hence its class; then each tile's rows are put
    // in the order  general | truth table | always deferred | booleanity  (stable within a class: the shape order) and the arrays are
    // built again.  The stream kernel deals rows to waves by position: the general rows, whose verdicts are the expensive ones, are
    // then always waves 0 and 1's, and the general words and the outside wires, which it deals to other waves, never meet them.
    build_tiles();
    is_bit = in_run;
    is_bit[0] = 1;                                         // the constant wire: 1 in every witness (the kernels check it)
    for (const Row &r : rows) { uint32_t w = 0; if (boolean_wire(r, &w)) is_bit[w] = 1; }
    std::vector<uint8_t> cls(m, 3);
    for (size_t r = 0; r < trow_k.size(); r++)
      cls[trow_k[r]] = (uint8_t)row_class(rows[trow_k[r]], (lrows[4 * r + 1] & 0x3FFFFFFFu) + lrows[4 * r + 2] + (lrows[4 * r + 1] >> 31 ? 0u : lrows[4 * r + 3]), nullptr);
    for (uint32_t t = 0; t < ntiles; t++)
      std::stable_sort(tile_rows[t].begin(), tile_rows[t].end(), [&](uint32_t a, uint32_t b) { return cls[a] < cls[b]; });
    build_tiles();
  }
  // ---- the STREAM program (b3w_r1cs_stream_kernel): the same tiles and the same row order, every row in one of four classes
  //   B  booleanity                 (descriptor only, as in the lean rows)
  //   T  TRUTH TABLE: a row over at most five distinct wires each of which is expected to hold a bit (a booleanity row of its own,
  //      a place in a bit run, or the constant wire): whether it holds is a function of at most five bits, tabulated HERE with exact field arithmetic — any
  //      coefficients — and looked up by the kernel from the elements' bits (87 % of the non-booleanity rows of blake3_compression:
  //      the XOR gates (2a)(b) = a + b - out).  An element that turns out to be no bit defers the row to the field arithmetic.
  //   D  always deferred            (a coefficient that is no small integer, or more than 256 words)
  //   G  general: its lean words go to a flat list of entries evaluated one ENTRY per lane (contributions added into per-row sums
  //      in LDS), instead of one row per lane walking its words
  std::vector<uint32_t> srows, sgdesc(4 * (size_t)ntiles), sgwords, sgmeta;
  uint32_t max_g_words = 0, max_g_rows = 0;
  // per tile and group of 64 LDS elements: the elements a booleanity or truth-table row of the tile takes for bits.  While none of
  // them is anything else (and wire 0 is 1) the kernel decides those rows without looking at them one by one.
  const uint32_t smask_groups = (T + ((max_ext + 32u) & ~31u) + 63u) >> 6;
  std::vector<unsigned long long> smask((size_t)ntiles * smask_groups, 0ull);
  if (tiled) {
    uint32_t inv = P[0];                                   // Newton: p^-1 mod 2^32
    for (int i = 0; i < 5; i++) inv *= 2u - P[0] * inv;
    inv = 0u - inv;
    const uint32_t one[8] = {1, 0, 0, 0, 0, 0, 0, 0};
    // (is_bit — wires expected to hold bits: a booleanity row of their own, or a place in a bit run (circom's XOR gate constrains
    // its output only through the gate itself; the outputs are then recomposed into words).  A wrong guess costs time, not
    // correctness: an element that is no bit defers its truth-table rows to the field arithmetic.)
    std::map<std::vector<uint32_t>, uint32_t> table_of;    // (wires' positions, coefficient ids, part lengths) -> truth table
    std::vector<uint32_t> run_w, run_m, term_w, term_m;
    for (uint32_t t = 0; t < ntiles; t++) {
      const uint32_t gw0 = (uint32_t)sgwords.size();
      uint32_t ng = 0;
      auto lds_index = [&](uint32_t w) { return w / T == t ? w - t * T : T + tile_ext[t].at(w); };
      for (uint32_t k : tile_rows[t]) {
        const Row &r = rows[k];
        const uint32_t nt = r.na + r.nb + r.nc;
        uint32_t bw = 0;
        auto must_be_bit = [&](uint32_t idx) { smask[(size_t)t * smask_groups + (idx >> 6)] |= 1ull << (idx & 63u); };
        if (boolean_wire(r, &bw)) { must_be_bit(lds_index(bw)); srows.insert(srows.end(), {0u, 0x80000000u | 1u, 2u, lds_index(bw)}); continue; }
        // the row's lean words (what emit_part made of it: the lean rows are in the same order)
        const size_t lr = srows.size();                     // (= 4 * this row's number in lrows)
        const uint32_t off = lrows[lr], n3[3] = {lrows[lr + 1] & 0x3FFFFFFFu, lrows[lr + 2], lrows[lr + 3]};
        const uint32_t nw = n3[0] + n3[1] + n3[2];
        std::vector<uint32_t> W;
        const int cls = row_class(r, nw, &W);
        if (cls == 1) {
          std::vector<uint32_t> key = {r.na, r.nb, r.nc};
          for (uint32_t x = 0; x < nt; x++) {
            key.push_back((uint32_t)(std::find(W.begin(), W.end(), wires[r.off + x]) - W.begin()));
            key.push_back(cids[r.off + x]);
          }
          auto it = table_of.find(key);
          if (it == table_of.end()) {
            uint32_t table = 0;
            for (uint32_t a = 0; a < (1u << W.size()); a++) {
              uint32_t sum[3][8];
              memset(sum, 0, sizeof sum);
              uint32_t q = 0;
              const uint32_t plen[3] = {r.na, r.nb, r.nc};
              for (int part = 0; part < 3; part++)
                for (uint32_t x = 0; x < plen[part]; x++, q++)
                  if ((a >> key[3 + 2 * q]) & 1u) u256_add_mod(sum[part], coefs[cids[r.off + q]].data(), P);
              uint32_t ab[8], cr[8];
              mont_mul_host(ab, sum[0], sum[1], P, inv);    // A * B / R
              mont_mul_host(cr, sum[2], one, P, inv);       // C / R
              if (!memcmp(ab, cr, 32)) table |= 1u << a;
            }
            it = table_of.emplace(key, table).first;
          }
          uint32_t idx[5] = {0, 0, 0, 0, 0};
          for (size_t j = 0; j < W.size(); j++) { idx[j] = lds_index(W[j]); must_be_bit(idx[j]); }
          srows.insert(srows.end(), {idx[0] | idx[1] << 16, 0x20000000u | (uint32_t)W.size() << 16 | idx[4], idx[2] | idx[3] << 16, it->second});
          continue;
        }
        // a bit run is packed as length (7 bits) | shift (6 bits) and the kernel shifts a run's sum inside 64 bits: a run the
        // fields or that arithmetic cannot hold (none in these systems: runs are cut at 64 bits and small coefficients stay below
        // 2^62) sends its row to the field arithmetic instead of overflowing the encoding
        bool runs_fit = true;
        for (uint32_t qq = off; qq < off + nw && runs_fit; qq++)
          if ((lterms[qq] >> 16) == 0xFFFFu && qq + 1 < off + nw) {
            const uint32_t w1 = lterms[++qq], len = w1 & 0xFFu, sh = (w1 >> 8) & 0xFFu;
            runs_fit = len >= 1 && len <= 64 && sh + len <= 62;
          }
        if (cls == 2 || !runs_fit || ng >= 0xFFFFFFu) { srows.insert(srows.end(), {0u, 0x40000000u, 0u, 0u}); continue; }
        // entries {word, meta}: a TERM is  element | coefficient id << 16,  meta = part | row << 8;  a BIT RUN (two lean words) is
        // one entry  first element | length << 16 | shift << 23 | negative << 29,  meta = part | 8 | row << 8.  The tile's runs stand
        // first, padded with null entries (meta = 4) to a multiple of 64, then its terms: every chunk of 64 entries a wave takes is
        // of ONE kind (the kernel runs one kind's code per chunk, not both under lane masks).
        uint32_t q = off;
        for (uint32_t part = 0; part < 3; part++)
          for (uint32_t x = 0; x < n3[part]; x++, q++) {
            const uint32_t w = lterms[q];
            if ((w >> 16) == 0xFFFFu) {
              const uint32_t w1 = lterms[++q];               // length | shift << 8 | negative << 16
              x++;
              run_w.push_back((w & 0xFFFFu) | (w1 & 0xFFu) << 16 | ((w1 >> 8) & 0xFFu) << 23 | ((w1 >> 16) & 1u) << 29);
              run_m.push_back(part | 8u | ng << 8);
            } else {
              term_w.push_back(w);
              term_m.push_back(part | ng << 8);
            }
          }
        srows.insert(srows.end(), {ng, 0x10000000u, 0u, 0u});
        ng++;
      }
      while (run_w.size() & 63u) { run_w.push_back(0u); run_m.push_back(4u); }
      sgdesc[4 * t] = gw0; sgdesc[4 * t + 1] = (uint32_t)(run_w.size() + term_w.size()); sgdesc[4 * t + 2] = ng; sgdesc[4 * t + 3] = (uint32_t)run_w.size();
      sgwords.insert(sgwords.end(), run_w.begin(), run_w.end()); sgwords.insert(sgwords.end(), term_w.begin(), term_w.end());
      sgmeta.insert(sgmeta.end(), run_m.begin(), run_m.end()); sgmeta.insert(sgmeta.end(), term_m.begin(), term_m.end());
      run_w.clear(); run_m.clear(); term_w.clear(); term_m.clear();
      max_g_words = std::max(max_g_words, sgdesc[4 * t + 1]);
      max_g_rows = std::max(max_g_rows, ng);
    }
    sgwords.push_back(0); sgmeta.push_back(4u);            // (a spare null entry)
  }
  // what a (body, tile) unit costs the stream kernel, relative: a fixed part (barriers, descriptors), the tile's bytes, its
  // general words (a tile of blake3_compression with 170 words: 3.9 us; + 2 ns per word) — the kernel deals units
  // to its persistent workgroups by cost, not by count: tile 23 of the circomkit nova build has 1 246 general words, seven times
  // the others', and a workgroup that drew only such units would finish half again as late as the rest
  std::vector<unsigned long long> scost(ntiles + 1, 0ull);
  if (tiled)
    for (uint32_t t = 0; t < ntiles; t++) {
      const uint32_t n_local = std::min<uint32_t>(T, nwires - t * T);
      // (calibrated on the three tilings, profiles/r03/r1cs_cost_calibration.log: fixed 1 500, bytes 400 per full tile, 1 per word)
      scost[t + 1] = scost[t] + 1500u + (400u * n_local) / T + sgdesc[4 * t + 1];
    }
  // ---- the WALK program (b3w_r1cs_walk_kernel; b3w_r1cs_host.h).  Rows go to the tile of their highest wire; what they mention
  // in earlier tiles is exported by those tiles.  Row order inside a tile: general | truth-table rows, grouped into RUNS |
  // always deferred | booleanity.  A RUN = up to 32 truth-table rows with the same table whose operands each either stay put or
  // advance by one element from row to row (the 32 XOR gates of a word: a_i, b_i, out_i) — one lane decides the whole run from
  // the bit-packed elements.
  if (tiled) {
    uint32_t inv = P[0];
    for (int i = 0; i < 5; i++) inv *= 2u - P[0] * inv;
    inv = 0u - inv;
    const uint32_t one[8] = {1, 0, 0, 0, 0, 0, 0, 0};
    std::vector<uint32_t> home(m);
    std::vector<std::vector<uint32_t>> wrows_of(ntiles);
    std::vector<uint8_t> exported(nwires, 0);
    for (uint32_t k = 0; k < m; k++) {
      const Row &r = rows[k];
      uint32_t hi = 0;
      for (uint32_t x = 0; x < r.na + r.nb + r.nc; x++) hi = std::max(hi, wires[r.off + x]);
      home[k] = hi / T;
      wrows_of[home[k]].push_back(k);
      for (uint32_t x = 0; x < r.na + r.nb + r.nc; x++) if (wires[r.off + x] / T != home[k]) exported[wires[r.off + x]] = 1;
    }
    std::vector<uint32_t> slot_of(nwires, 0), slot0(ntiles, 0), exp_off(ntiles, 0), exp_n(ntiles, 0);
    std::vector<uint16_t> wexp;
    // The kernel's UNITS per body: a tile, or — where a tile has more general rows than B3W_WALK_SPLIT_GEN — the tile several times
    // over, each unit with a share of those rows (the first one also with the tile's exports, runs and masks' rows).  The row sums
    // in LDS are sized by the largest unit (96 bytes per general row, twice): one tile of 425 such rows (the circomkit nova build:
    // its last full tile) would otherwise push a workgroup past half a CU's LDS and the whole system off the walk kernel.
    std::vector<uint32_t> wtile;                            // B3W_WT_WORDS per unit
    uint32_t slots = 0, wmax_exp = 0;
    for (uint32_t t = 0; t < ntiles; t++) {
      slot0[t] = slots;
      exp_off[t] = (uint32_t)wexp.size();
      uint32_t ne = 0;
      for (uint32_t w = t * T; w < std::min(nwires, (t + 1) * T); w++)
        if (exported[w]) { slot_of[w] = slots + ne; wexp.push_back((uint16_t)(w - t * T)); ne++; }
      exp_n[t] = ne;
      wmax_exp = std::max(wmax_exp, ne);
      slots += (ne + 63u) & ~63u;
    }
    // (a term word is element | coefficient id << 16 with bit 31 = "shift term": ids from 0x8000 on would read as shifts.  Such a
    // system would not fit the kernel's LDS either — its coefficient table alone is 256 KB — but that is a budget, not a guarantee)
    bool walk = slots <= B3W_WALK_MAX_EXP_SLOTS && coefs.size() < 0x8000;
    auto widx = [&](uint32_t w, uint32_t t) { return w / T == t ? w - t * T : T + slot_of[w]; };
    std::vector<uint8_t> mustbit(nwires, 0);
    std::vector<uint32_t> wruns, wrun_row, went_w, went_m, wrow_k, wrow_id, wtiles4, unit_tile;
    uint32_t wmax_gen = 0, wmax_ent = 0, wmax_runs = 0, wmax_rows = 0;
    std::map<std::vector<uint32_t>, uint32_t> wtable_of;    // (part lengths, per term: operand position, coefficient id) -> truth table
    std::vector<std::vector<uint32_t>> static_rows;         // per unit: always-deferred rows, as positions in the unit
    for (uint32_t t = 0; t < ntiles && walk; t++) {
      // this tile's rows by class, with what each class needs
      struct TRow { uint32_t k, table, nops; uint32_t idx[5]; uint32_t keyid; };
      struct GRow { uint32_t k; std::vector<uint32_t> ew, em; };   // entries without the row number (filled in below)
      std::vector<GRow> gen;
      std::vector<TRow> tts;
      std::vector<uint32_t> defs, bools;
      std::map<std::vector<uint32_t>, uint32_t> keyids;
      for (uint32_t k : wrows_of[t]) {
        const Row &r = rows[k];
        uint32_t bw = 0;
        if (boolean_wire(r, &bw)) { bools.push_back(k); mustbit[bw] = 1; continue; }
        // the row's entries (what the general road would take): per part, terms sorted by element, bit runs folded
        GRow g{k, {}, {}};
        const uint32_t plen[3] = {r.na, r.nb, r.nc};
        uint32_t q = 0;
        std::vector<std::vector<uint32_t>> run_wires;
        for (uint32_t part = 0; part < 3; part++) {
          std::vector<std::array<uint32_t, 3>> terms;          // element, coefficient id, wire
          for (uint32_t x = 0; x < plen[part]; x++, q++) terms.push_back({widx(wires[r.off + q], t), cids[r.off + q], wires[r.off + q]});
          std::stable_sort(terms.begin(), terms.end(), [](const std::array<uint32_t, 3> &a, const std::array<uint32_t, 3> &b) { return a[0] < b[0]; });
          for (size_t i = 0; i < terms.size();) {
            bool neg = false; uint32_t k0 = 0;
            size_t j = i + 1;
            if (pow2((uint16_t)terms[i][1], neg, k0)) {
              while (j < terms.size() && j - i < 64 && terms[j][0] == terms[j - 1][0] + 1) {
                bool ng = false; uint32_t kk = 0;
                if (!pow2((uint16_t)terms[j][1], ng, kk) || ng != neg || kk != k0 + (uint32_t)(j - i)) break;
                j++;
              }
            }
            // (a run never crosses from the local elements into the export area: the two are different arrays in LDS)
            while (j - i >= 4 && (terms[i][0] < T) != (terms[j - 1][0] < T)) j--;
            if (j - i >= 4 && k0 + (uint32_t)(j - i) <= 62) {
              g.ew.push_back(terms[i][0] | (uint32_t)(j - i) << 16 | k0 << 23 | (neg ? 1u << 29 : 0u));
              g.em.push_back(part | 8u);
              std::vector<uint32_t> ws;
              for (size_t x = i; x < j; x++) ws.push_back(terms[x][2]);
              run_wires.push_back(ws);
              i = j;
            } else {
              // a term: element | coefficient id << 16 — or, for a coefficient +-2^k (98 % of them), element | k << 16 | negative << 22
              // | 1 << 31: the kernel shifts instead of multiplying and needs no coefficient from its table
              bool ng = false; uint32_t kk = 0;
              static const bool shift_terms = !(getenv("B3W_WALK_SHIFT_TERMS") && !strcmp(getenv("B3W_WALK_SHIFT_TERMS"), "0"));      // (0: measurements)
              if (shift_terms && pow2((uint16_t)terms[i][1], ng, kk)) g.ew.push_back(terms[i][0] | kk << 16 | (ng ? 1u << 22 : 0u) | 1u << 31);
              else g.ew.push_back(terms[i][0] | terms[i][1] << 16);
              g.em.push_back(part);
              i++;
            }
          }
        }
        std::vector<uint32_t> W;
        const int cls = row_class(r, (uint32_t)g.ew.size(), &W);
        if (cls == 1) {
          const uint32_t nt = r.na + r.nb + r.nc;
          std::vector<uint32_t> key = {r.na, r.nb, r.nc};
          for (uint32_t x = 0; x < nt; x++) {
            key.push_back((uint32_t)(std::find(W.begin(), W.end(), wires[r.off + x]) - W.begin()));
            key.push_back(cids[r.off + x]);
          }
          auto it = wtable_of.find(key);
          if (it == wtable_of.end()) {
            uint32_t table = 0;
            for (uint32_t a = 0; a < (1u << W.size()); a++) {
              uint32_t sum[3][8];
              memset(sum, 0, sizeof sum);
              uint32_t qq = 0;
              for (int part = 0; part < 3; part++)
                for (uint32_t x = 0; x < plen[part]; x++, qq++)
                  if ((a >> key[3 + 2 * qq]) & 1u) u256_add_mod(sum[part], coefs[cids[r.off + qq]].data(), P);
              uint32_t ab[8], cr[8];
              mont_mul_host(ab, sum[0], sum[1], P, inv);
              mont_mul_host(cr, sum[2], one, P, inv);
              if (!memcmp(ab, cr, 32)) table |= 1u << a;
            }
            it = wtable_of.emplace(key, table).first;
          }
          TRow tr{k, it->second, (uint32_t)W.size(), {0, 0, 0, 0, 0}, 0};
          for (size_t j = 0; j < W.size(); j++) { tr.idx[j] = widx(W[j], t); mustbit[W[j]] = 1; }
          auto kid = keyids.find(key);
          if (kid == keyids.end()) kid = keyids.emplace(key, (uint32_t)keyids.size()).first;
          tr.keyid = kid->second;
          tts.push_back(tr);
          continue;
        }
        if (cls == 2) { defs.push_back(k); continue; }
        for (const auto &ws : run_wires) for (uint32_t w : ws) mustbit[w] = 1;
        gen.push_back(std::move(g));
      }
      // truth-table rows into runs
      std::stable_sort(tts.begin(), tts.end(), [](const TRow &a, const TRow &b) {
        if (a.keyid != b.keyid) return a.keyid < b.keyid;
        for (int j = 0; j < 5; j++) if (a.idx[j] != b.idx[j]) return a.idx[j] < b.idx[j];
        return false;
      });
      // (B3W_WALK_SPLIT_GEN in the environment: another threshold, for measurements — at most the compiled one, which sizes nothing)
      static const uint32_t split_gen = getenv("B3W_WALK_SPLIT_GEN") ? std::min<uint32_t>(B3W_WALK_SPLIT_GEN, std::max(16, atoi(getenv("B3W_WALK_SPLIT_GEN")))) : B3W_WALK_SPLIT_GEN;
      const uint32_t nparts = std::max<uint32_t>(1u, ((uint32_t)gen.size() + split_gen - 1u) / split_gen);
      for (uint32_t part = 0; part < nparts && walk; part++) {
      const size_t g_lo = gen.size() * part / nparts, g_hi = gen.size() * (part + 1u) / nparts;
      const size_t u = unit_tile.size();                    // this unit
      unit_tile.push_back(t);
      wtile.resize((u + 1) * (size_t)B3W_WT_WORDS, 0u);
      wtiles4.resize(4 * (u + 1), 0u);
      static_rows.emplace_back();
      wtile[(size_t)B3W_WT_WORDS * u + B3W_WT_NLOCAL] = std::min<uint32_t>(T, nwires - t * T);
      wtile[(size_t)B3W_WT_WORDS * u + B3W_WT_SRC] = t;
      wtile[(size_t)B3W_WT_WORDS * u + B3W_WT_EXP_OFF] = exp_off[t];
      wtile[(size_t)B3W_WT_WORDS * u + B3W_WT_EXP_N] = part ? 0u : exp_n[t];
      wtile[(size_t)B3W_WT_WORDS * u + B3W_WT_EXP_SLOT0] = slot0[t];
      const uint32_t row0 = (uint32_t)wrow_k.size();
      wtile[(size_t)B3W_WT_WORDS * u + B3W_WT_ROW0] = row0;
      wtile[(size_t)B3W_WT_WORDS * u + B3W_WT_GEN_N] = (uint32_t)(g_hi - g_lo);
      wtile[(size_t)B3W_WT_WORDS * u + B3W_WT_ENT_OFF] = (uint32_t)went_w.size();
      {
        std::vector<uint32_t> rw, rm, tw, tm, ow, om;        // bit runs | terms with a power-of-two coefficient | other terms
        for (size_t gi = g_lo; gi < g_hi; gi++) {
          for (size_t e = 0; e < gen[gi].ew.size(); e++) {
            const uint32_t mt = gen[gi].em[e] | (uint32_t)(gi - g_lo) << 8;
            if (gen[gi].em[e] & 8u) { rw.push_back(gen[gi].ew[e]); rm.push_back(mt); }
            else if (gen[gi].ew[e] >> 31) { tw.push_back(gen[gi].ew[e]); tm.push_back(mt); }
            else { ow.push_back(gen[gi].ew[e]); om.push_back(mt); }
          }
          wrow_k.push_back(gen[gi].k); wrow_id.push_back(rows[gen[gi].k].id);
        }
        // (the other terms last: a chunk of 64 whose terms all shift takes the kernel's short road — one mixed chunk a unit at most)
        tw.insert(tw.end(), ow.begin(), ow.end()); tm.insert(tm.end(), om.begin(), om.end());
        while (rw.size() & 63u) { rw.push_back(0u); rm.push_back(4u); }
        wtile[(size_t)B3W_WT_WORDS * u + B3W_WT_ENT_RUNS] = (uint32_t)rw.size();
        wtile[(size_t)B3W_WT_WORDS * u + B3W_WT_ENT_N] = (uint32_t)(rw.size() + tw.size());
        went_w.insert(went_w.end(), rw.begin(), rw.end()); went_w.insert(went_w.end(), tw.begin(), tw.end());
        went_m.insert(went_m.end(), rm.begin(), rm.end()); went_m.insert(went_m.end(), tm.begin(), tm.end());
        wmax_ent = std::max<uint32_t>(wmax_ent, (uint32_t)(rw.size() + tw.size()));
      }
      wtile[(size_t)B3W_WT_WORDS * u + B3W_WT_RUN_OFF] = (uint32_t)(wruns.size() / 4);
      for (size_t i = 0; i < tts.size() && part == 0u;) {
        uint32_t stride[5] = {0, 0, 0, 0, 0};
        size_t j = i + 1;
        if (j < tts.size() && tts[j].keyid == tts[i].keyid) {
          bool ok = true;
          for (uint32_t o = 0; o < tts[i].nops && ok; o++) {
            const uint32_t d = tts[j].idx[o] - tts[i].idx[o];
            ok = d <= 1;
            stride[o] = d;
          }
          if (ok) {
            j++;
            while (j < tts.size() && j - i < 32 && tts[j].keyid == tts[i].keyid) {
              bool same = true;
              for (uint32_t o = 0; o < tts[i].nops && same; o++) same = tts[j].idx[o] == tts[i].idx[o] + stride[o] * (uint32_t)(j - i);
              if (!same) break;
              j++;
            }
          } else { for (uint32_t &x : stride) x = 0; }
        }
        // an advancing operand must stay inside its array (local elements or export area) and inside what one funnel read covers
        for (uint32_t o = 0; o < tts[i].nops; o++)
          if (stride[o]) while (j - i > 1 && ((tts[i].idx[o] < T) != (tts[i].idx[o] + (uint32_t)(j - i) - 1 < T))) j--;
        if (j - i == 1) for (uint32_t &x : stride) x = 0;
        const TRow &a = tts[i];
        uint32_t sb = 0;
        for (uint32_t o = 0; o < 5; o++) sb |= (stride[o] ? 1u : 0u) << o;
        wruns.push_back(a.table);
        wruns.push_back(a.idx[0] | a.idx[1] << 16);
        wruns.push_back(a.idx[2] | a.idx[3] << 16);
        wruns.push_back(a.idx[4] | (uint32_t)(j - i - 1) << 16 | a.nops << 21 | sb << 24);
        wrun_row.push_back((uint32_t)wrow_k.size());
        for (size_t x = i; x < j; x++) { wrow_k.push_back(tts[x].k); wrow_id.push_back(rows[tts[x].k].id); }
        i = j;
      }
      wtile[(size_t)B3W_WT_WORDS * u + B3W_WT_RUN_N] = (uint32_t)(wruns.size() / 4) - wtile[(size_t)B3W_WT_WORDS * u + B3W_WT_RUN_OFF];
      wmax_runs = std::max(wmax_runs, wtile[(size_t)B3W_WT_WORDS * u + B3W_WT_RUN_N]);
      if (part == 0u) {
        for (uint32_t k : defs) { static_rows[u].push_back((uint32_t)wrow_k.size() - row0); wrow_k.push_back(k); wrow_id.push_back(rows[k].id); }
        for (uint32_t k : bools) { wrow_k.push_back(k); wrow_id.push_back(rows[k].id); }
      }
      const uint32_t nrows = (uint32_t)wrow_k.size() - row0;
      wtile[(size_t)B3W_WT_WORDS * u + B3W_WT_NROWS] = nrows;
      wtiles4[4 * u] = row0; wtiles4[4 * u + 1] = nrows;
      wmax_gen = std::max<uint32_t>(wmax_gen, (uint32_t)(g_hi - g_lo));
      wmax_rows = std::max(wmax_rows, nrows);
      if (g_hi - g_lo > B3W_WALK_MAX_GEN || wmax_ent > B3W_WALK_MAX_ENT || nrows > 4096u || unit_tile.size() > B3W_WALK_MAX_UNITS) { if (getenv("B3W_WALK_DEBUG")) fprintf(stderr, "walk: tile %u gen %zu ent %u rows %u\n", t, g_hi - g_lo, wmax_ent, nrows); walk = false; }
      }
    }
    if (walk) {
      const uint32_t sw = (wmax_rows + 63u) / 64u, nunits = (uint32_t)unit_tile.size();
      std::vector<unsigned long long> wstatic((size_t)nunits * sw, 0ull), wmask((size_t)nunits * 16u, 0ull);
      for (uint32_t u = 0; u < nunits; u++) {               // (every unit of a tile carries the tile's must-be-bit mask: the same anomaly, seen again)
        const uint32_t t = unit_tile[u];
        for (uint32_t pos : static_rows[u]) wstatic[(size_t)u * sw + (pos >> 6)] |= 1ull << (pos & 63u);
        for (uint32_t w = t * T; w < std::min(nwires, (t + 1) * T); w++)
          if (mustbit[w]) wmask[(size_t)u * 16u + ((w - t * T) >> 6)] |= 1ull << ((w - t * T) & 63u);
      }
      // (a wave reads whole chunks of 64: a lane may read up to 63 entries, descriptors or exports behind its tile's last)
      for (int pad = 0; pad < 64; pad++) { wruns.insert(wruns.end(), {0u, 0u, 0u, 0u}); went_w.push_back(0u); went_m.push_back(4u); wexp.push_back(0); }
      uint32_t nlin = 0;
      for (uint32_t k = 0; k < m; k++) if (rows[k].na == 0 || rows[k].nb == 0) nlin++;
      H->wlinear_rows = nlin;
      H->walk = true; H->wunits = nunits; H->wexp_slots = slots; H->wmax_gen = wmax_gen; H->wmax_ent = wmax_ent; H->wmax_exp = wmax_exp; H->wmax_runs = wmax_runs;
      H->wmax_rows = wmax_rows; H->wstatic_words = sw;
      H->wtile = std::move(wtile); H->wmask = std::move(wmask); H->wexp = std::move(wexp); H->wruns = std::move(wruns);
      H->wrun_row = std::move(wrun_row); H->went_w = std::move(went_w); H->went_m = std::move(went_m); H->wrow_k = std::move(wrow_k);
      H->wrow_id = std::move(wrow_id); H->wtiles4 = std::move(wtiles4); H->wstatic = std::move(wstatic);
      {
        std::vector<uint32_t> &sk = H->wstatic_list, &sid = H->wstatic_ids, srows;
        sk.clear(); sid.clear();
        for (uint32_t u = 0; u < nunits; u++)
          for (uint32_t pos = 0; pos < H->wtiles4[4 * (size_t)u + 1]; pos++)
            if ((H->wstatic[(size_t)u * sw + (pos >> 6)] >> (pos & 63u)) & 1ull) {
              srows.push_back(H->wrow_k[H->wtiles4[4 * (size_t)u] + pos]);
              sid.push_back(H->wrow_id[H->wtiles4[4 * (size_t)u] + pos]);
            }
        sk.assign(4 * srows.size(), 0u);
        for (size_t i = 0; i < srows.size(); i++) {
          const Row &r = rows[srows[i]];
          const uint32_t plen[3] = {r.na, r.nb, r.nc};
          const size_t first = sk.size() / 2;
          std::map<uint64_t, size_t> seen;                   // (wire, coefficient) -> its pair, while no part has it twice
          uint32_t q = r.off;
          for (uint32_t part = 0; part < 3; part++)
            for (uint32_t x = 0; x < plen[part]; x++, q++) {
              const uint64_t key = (uint64_t)wires[q] << 16 | cids[q];
              auto it = seen.find(key);
              if (it != seen.end() && !((sk[it->second + 1] >> (16 + part)) & 1u)) { sk[it->second + 1] |= 1u << (16 + part); continue; }
              seen[key] = sk.size();
              sk.push_back(wires[q]);
              sk.push_back((uint32_t)cids[q] | 1u << (16 + part));
            }
          sk[4 * i] = (uint32_t)first; sk[4 * i + 1] = (uint32_t)(sk.size() / 2 - first);
          sk[4 * i + 2] = r.na == 0 || r.nb == 0 ? 1u : 0u; sk[4 * i + 3] = r.nc ? 1u : 0u;
        }
      }
    }
  }
  std::vector<uint32_t> coefR(16 * coefs.size());           // per coefficient: plain, then Montgomery form
  for (size_t i = 0; i < coefs.size(); i++) {
    memcpy(&coefR[16 * i], coefs[i].data(), 32);
    memcpy(&coefR[16 * i + 8], coefs[i].data(), 32);
    to_montgomery_host(&coefR[16 * i + 8], P);
  }
  H->m = m; H->nwires = nwires; H->npubout = npubout; H->npubin = npubin; H->nprvin = nprvin; H->nterms = wires.size();
  memcpy(H->field.p, P, 32);
  uint32_t r2[8] = {1, 0, 0, 0, 0, 0, 0, 0};
  to_montgomery_host(r2, P);
  to_montgomery_host(r2, P);                              // 2^512 mod p
  memcpy(H->field.r2, r2, 32);
  uint32_t inv = P[0];                                    // Newton: p^-1 mod 2^32
  for (int i = 0; i < 5; i++) inv *= 2u - P[0] * inv;
  H->field.inv = 0u - inv;
  H->tiled = tiled; H->ntiles = ntiles; H->max_ext = max_ext; H->max_tile_terms = max_tile_terms; H->max_tile_rows = max_tile_rows;
  H->max_lean_terms = max_lean_terms; H->ncoef = (uint32_t)coefs.size();
  H->rowdesc = std::move(rowdesc); H->row_id = std::move(row_id); H->wires = std::move(wires); H->cids = std::move(cids);
  H->coefR = std::move(coefR); H->coef_small = std::move(coef_small);
  H->tdesc = std::move(tdesc); H->ttdesc = std::move(ttdesc); H->text = std::move(text); H->trows = std::move(trows);
  H->trow_id = std::move(trow_id); H->trow_k = std::move(trow_k); H->tterms = std::move(tterms);
  H->ltdesc = std::move(ltdesc); H->lrows = std::move(lrows); H->lterms = std::move(lterms);
  H->srows = std::move(srows); H->sgdesc = std::move(sgdesc); H->sgwords = std::move(sgwords); H->sgmeta = std::move(sgmeta);
  H->max_g_words = max_g_words; H->max_g_rows = max_g_rows;
  H->smask = std::move(smask); H->smask_groups = smask_groups; H->scost = std::move(scost);
  return true;
}
