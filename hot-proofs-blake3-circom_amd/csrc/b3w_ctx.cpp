// b3w_ctx.cpp — C-ABI of libb3wit.so (include/b3wit.h), part 1: circuits, slot tables, the context, the batch launch entry points,
// one witness through the calculator surface.  Host code only; every witness is computed by the HIP kernels in b3w_kernels.hip.
// There is no CPU evaluation path in this library: without a HIP device b3w_create fails with B3W_E_NO_DEVICE.
#include "b3w_internal.h"
#include "b3w_layout_tables.inc"
#include "b3w_trace_tables.inc"

namespace {

// ------------------------------------------------------------------ sha256 (FIPS 180-4), for b3w_identify_wasm
struct Sha256 {
  uint32_t h[8];
  uint8_t buf[64];
  uint64_t len = 0;
  size_t fill = 0;
  Sha256() {
    static const uint32_t init[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    memcpy(h, init, sizeof h);
  }
  static uint32_t ror(uint32_t x, int r) { return (x >> r) | (x << (32 - r)); }
  void block(const uint8_t *p) {
    static const uint32_t K[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
        0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
        0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
        0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
        0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
        0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
        0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
    uint32_t w[64];
    for (int i = 0; i < 16; i++) w[i] = (uint32_t)p[4 * i] << 24 | (uint32_t)p[4 * i + 1] << 16 | (uint32_t)p[4 * i + 2] << 8 | p[4 * i + 3];
    for (int i = 16; i < 64; i++) {
      uint32_t s0 = ror(w[i - 15], 7) ^ ror(w[i - 15], 18) ^ (w[i - 15] >> 3);
      uint32_t s1 = ror(w[i - 2], 17) ^ ror(w[i - 2], 19) ^ (w[i - 2] >> 10);
      w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; i++) {
      uint32_t t1 = hh + (ror(e, 6) ^ ror(e, 11) ^ ror(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
      uint32_t t2 = (ror(a, 2) ^ ror(a, 13) ^ ror(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
      hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
  }
  void update(const uint8_t *p, size_t n) {
    len += n;
    while (n) {
      size_t k = 64 - fill < n ? 64 - fill : n;
      memcpy(buf + fill, p, k);
      fill += k; p += k; n -= k;
      if (fill == 64) { block(buf); fill = 0; }
    }
  }
  void final(uint8_t out[32]) {
    uint64_t bits = len * 8;
    uint8_t pad = 0x80;
    update(&pad, 1);
    uint8_t z = 0;
    while (fill != 56) update(&z, 1);
    uint8_t lb[8];
    for (int i = 0; i < 8; i++) lb[i] = (uint8_t)(bits >> (56 - 8 * i));
    update(lb, 8);
    for (int i = 0; i < 8; i++) { out[4 * i] = h[i] >> 24; out[4 * i + 1] = h[i] >> 16; out[4 * i + 2] = h[i] >> 8; out[4 * i + 3] = h[i]; }
  }
};

// sha256 of the reference's committed circuit binaries
const char *const WASM_SHA256[4] = {
    "6faf23ddfd697bbb7e8e922577589c2c06486258968a5a14f96fb5a16091b142",   // blake3_compression.wasm
    "020bd11f289864c54c7d02cd05723dcf8323e31fa5c77d8700c618232685978e",   // build/blake3_nova_js/blake3_nova.wasm
    "b982f960ebbfcabe957fe13857ea47adfeee30e18fbe05474e9b982eab187f46",   // build/blake3_nova_pasta_js/blake3_nova_pasta.wasm
    "8d6317b72eab34d34e12dfd7bd310dce40f4190768669772f992a9510c441fca"};  // build/blake3_nova/.../blake3_nova.wasm (== circomkit "pasta")


// FNV-1a 64 of the signal name (witness_calculator.js:325-337)
uint64_t fnv1a64(const char *s) {
  uint64_t h = 0xCBF29CE484222325ull;
  for (; *s; ++s) { h ^= (uint8_t)*s; h *= 0x100000001B3ull; }
  return h;
}
const CircuitDesc CIRCUITS[4] = {
    {B3W_KIND_COMP, 24093, 28, 16, P_BN254, B3W_LAYOUT_0, B3W_LAYOUT_0_NRUNS, B3W_LDS_WORDS_COMP},
    {B3W_KIND_NOVA_O2, 23291, 32, 15, P_BN254, B3W_LAYOUT_1, B3W_LAYOUT_1_NRUNS, B3W_LDS_WORDS_NOVA_O2},
    {B3W_KIND_NOVA_O2, 23291, 32, 15, P_VESTA, B3W_LAYOUT_2, B3W_LAYOUT_2_NRUNS, B3W_LDS_WORDS_NOVA_O2},
    {B3W_KIND_NOVA_O1, 24614, 32, 15, P_BN254, B3W_LAYOUT_3, B3W_LAYOUT_3_NRUNS, B3W_LDS_WORDS_NOVA_O1},
};

// where an atom lives in the LDS image (b3w_atoms.h); returns false if this kind does not stage it
bool atom_lds(int kind, uint32_t atom, uint32_t *word, int *width /* 32, 64 or 256 */) {
  if (atom < B3W_A_HG) { *word = atom; *width = 32; return true; }
  if (atom < B3W_A_NV) {
    const uint32_t k = (atom - B3W_A_HG) / 8, j = (atom - B3W_A_HG) % 8;
    static const uint32_t off[8] = {0, 0, 2, 2, 4, 5, 6, 7};   // S1 A S3 C D2 DI B4 BI
    *word = B3W_LDS_HG + 8 * k + off[j];
    *width = (j == B3W_HG_S1 || j == B3W_HG_S3) ? 64 : 32;
    return true;
  }
  if (kind == B3W_KIND_COMP) return false;
  const uint32_t i = atom - B3W_A_NV;
  if (i == NV_CHUNK_IDX) { *word = B3W_LDS_CHUNK_IDX; *width = 64; return true; }
  if (i < NV_NARROW_COUNT) { *word = B3W_LDS_NV + i; *width = 32; return true; }
  if (i >= NV_COUNT) return false;
  uint32_t j = i - NV_NARROW_COUNT;                            // wide index in numbering order
  if (kind == B3W_KIND_NOVA_O2) {
    if (j < 3) { /* root/e0/e1 inv */ }
    else if (i >= NV_EQ_INV && i < NV_EQ_INV + 64) j = 3 + (i - NV_EQ_INV);
    else return false;
  }
  *word = B3W_LDS_WIDE + 8 * j;
  *width = 256;
  return true;
}

// 1/k mod p for 1 <= k < 2^32 (same closed form as the device routine in b3w_kernels.hip): with
// t = -p^-1 mod k, (p*t + 1)/k is exact and is the inverse.  Used to fill the device-side table of
// small inverses the nova kernels look IsZero arguments up in.
void inv_small_host(uint32_t k, const uint32_t P[8], uint32_t out[8]) {
  uint64_t r = 0;
  for (int i = 7; i >= 0; --i) r = ((r << 32) | P[i]) % k;
  int64_t x0 = 0, x1 = 1;
  uint64_t a = k, b = r;
  while (b > 1) {
    const uint64_t q = a / b, tt = a - q * b;
    a = b; b = tt;
    const int64_t tx = x0 - (int64_t)q * x1;
    x0 = x1; x1 = tx;
  }
  int64_t x = x1 % (int64_t)k;
  if (x < 0) x += k;
  const uint64_t t = ((uint64_t)k - (uint64_t)x) % k;
  uint32_t prod[9];
  uint64_t carry = 1;
  for (int i = 0; i < 8; ++i) {
    const uint64_t cur = (uint64_t)P[i] * t + carry;
    prod[i] = (uint32_t)cur;
    carry = cur >> 32;
  }
  prod[8] = (uint32_t)carry;
  uint64_t rem = 0;
  for (int i = 8; i >= 0; --i) {
    const uint64_t cur = (rem << 32) | prod[i];
    const uint64_t qd = cur / k;
    rem = cur - qd * k;
    if (i < 8) out[i] = (uint32_t)qd;
  }
}
// d_aux image: [0,8) prime limbs, [8] TABLE_N, [16 + 8k, +8) k^-1 mod p
std::vector<uint32_t> build_nova_aux(const uint64_t *prime) {
  std::vector<uint32_t> aux(16 + 8 * B3W_INV_TABLE_N, 0);
  memcpy(aux.data(), prime, 32);
  aux[8] = B3W_INV_TABLE_N;
  for (uint32_t k = 1; k < B3W_INV_TABLE_N; ++k) inv_small_host(k, aux.data(), aux.data() + 16 + 8 * k);
  return aux;
}

bool build_slot_table(const CircuitDesc &c, std::vector<uint32_t> &table, std::string &err) {
  const uint32_t padded = ((c.nwit + 31) / 32 + 8) * 32;   // + 8 groups: expand() prefetches ahead
  table.assign(padded, B3W_ENTRY(0, 31, B3W_MODE_BIT));        // padding: bit 31 of ONE = 0 (never stored anyway)
  std::vector<uint8_t> seen(c.nwit, 0);
  char msg[128];
  for (uint32_t r = 0; r < c.nruns; r++) {
    const b3w_layout_run &run = c.runs[r];
    for (uint32_t j = 0; j < run.len; j++) {
      const uint32_t slot = run.slot + j;
      if (slot >= c.nwit || seen[slot]) { err = "layout: bad or duplicate slot"; return false; }
      seen[slot] = 1;
      uint32_t word; int width;
      if (run.kind == 'W') {
        if (!atom_lds(c.kind, run.atom + j, &word, &width)) {
          snprintf(msg, sizeof msg, "layout: atom %u not staged for this circuit kind", run.atom + j);
          err = msg; return false;
        }
        table[slot] = B3W_ENTRY(word, 0, width == 32 ? B3W_MODE_W32 : width == 64 ? B3W_MODE_W64 : B3W_MODE_W256);
      } else {
        const uint32_t bit = run.bit0 + j;
        if (run.atom == B3W_A_NV + NV_CHUNK_IDX && bit == 64) {
          // Num2Bits(65).out[64] of chunk_idx = chunk_idx_low + 2^32*chunk_idx_high: always 0 for the
          // u32 inputs of the device path (kept as a slot only by the circomkit build)
          table[slot] = B3W_ENTRY(B3W_A_ONE, 31, B3W_MODE_BIT);
          continue;
        }
        if (!atom_lds(c.kind, run.atom, &word, &width) || (int)bit >= width || width == 256) {
          snprintf(msg, sizeof msg, "layout: bit %u of atom %u not expressible", bit, run.atom);
          err = msg; return false;
        }
        table[slot] = B3W_ENTRY(word + bit / 32, bit % 32, B3W_MODE_BIT);
      }
    }
  }
  for (uint32_t s = 0; s < c.nwit; s++) if (!seen[s]) { err = "layout: uncovered slot"; return false; }
  return true;
}

// slot -> atom | bit<<16 (0xFFFF = whole element) for the exact (field-element) kernel
bool build_exact_table(const CircuitDesc &c, std::vector<uint32_t> &table) {
  table.assign(c.nwit, 0);
  for (uint32_t r = 0; r < c.nruns; r++) {
    const b3w_layout_run &run = c.runs[r];
    for (uint32_t j = 0; j < run.len; j++) {
      const uint32_t slot = run.slot + j;
      if (slot >= c.nwit) return false;
      if (run.kind == 'W') table[slot] = (run.atom + j) | (0xFFFFu << 16);
      else table[slot] = run.atom | ((run.bit0 + j) << 16);
    }
  }
  return true;
}

// the reference WASM's own trace for this assert site, if it was tabulated (tools/probe_traces.py)
const char *reference_trace(int circuit, uint32_t site) {
  static const b3w_trace_entry *const T[4] = {B3W_TRACES_0, B3W_TRACES_1, B3W_TRACES_2, B3W_TRACES_3};
  static const uint32_t N[4] = {B3W_TRACES_0_N, B3W_TRACES_1_N, B3W_TRACES_2_N, B3W_TRACES_3_N};
  for (uint32_t i = 0; i < N[circuit]; i++) if (T[circuit][i].site == site) return T[circuit][i].text;
  return nullptr;
}

// VERIFY mode: record word j of a witness is read back from slot in_slots[j] of its body (the slot the
// layout gives the whole input atom: compression atoms 1..28, nova atoms NV+0..31 in record order)
bool build_input_slots(const CircuitDesc &c, std::vector<uint32_t> &slots) {
  slots.assign(c.nin, 0xFFFFFFFFu);
  const uint32_t first = c.kind == B3W_KIND_COMP ? B3W_A_H : B3W_A_NV;
  for (uint32_t r = 0; r < c.nruns; r++) {
    const b3w_layout_run &run = c.runs[r];
    if (run.kind != 'W') continue;
    for (uint32_t j = 0; j < run.len; j++) {
      const uint32_t atom = run.atom + j;
      if (atom >= first && atom < first + c.nin && slots[atom - first] == 0xFFFFFFFFu) slots[atom - first] = run.slot + j;
    }
  }
  for (uint32_t v : slots) if (v == 0xFFFFFFFFu) return false;
  return true;
}

const char *assert_site_text(uint32_t site, char *buf, size_t len) {
  const uint32_t code = site & 0xFF, r = (site >> 8) & 0xF, g = (site >> 12) & 0xF, hf = (site >> 16) & 1;
  switch (code) {
    case 1: snprintf(buf, len, "Error in template Bits34 line: 201 (rounds[%u].GS[%u].half%u.add1)\n", r, g, hf + 1); break;
    case 2: snprintf(buf, len, "Error in template ToBits line: 153 (rounds[%u].GS[%u].half%u.rxor2.tb)\n", r, g, hf + 1); break;
    case 3: snprintf(buf, len, "Error in template Bits33 line: 176 (rounds[%u].GS[%u].half%u.add3)\n", r, g, hf + 1); break;
    case 4: snprintf(buf, len, "Error in template ToBits line: 153 (rounds[%u].GS[%u].half%u.rxor4.tb)\n", r, g, hf + 1); break;
    case 5: snprintf(buf, len, "Error in template ToBits line: 153 (outXor[%u].tb_x)\n", g + 8 * hf); break;
    case 6: snprintf(buf, len, "Error in template ToBits line: 153 (outXor[%u].tb_y)\n", g + 8 * hf); break;
    case 10: snprintf(buf, len, "Error in template Num2Bits line: 38 (check_depth.check_parent.n2b)\n"); break;
    case 11: snprintf(buf, len, "Error in template Num2Bits line: 38 (check_depth.exceed_depth.lt.n2b)\n"); break;
    case 12: snprintf(buf, len, "Error in template Blake3NovaTreePath_CheckDepth line: 38\n"); break;
    case 13: snprintf(buf, len, "Error in template Num2Bits line: 38 (final_m.down_left_path.n2b)\n"); break;
    case 14: snprintf(buf, len, "Error in template Blake3GetDownLeftPath line: 77\n"); break;
    default: snprintf(buf, len, "assert site %u\n", site);
  }
  return buf;
}

}  // namespace

bool b3w_int_build_slot_table(const CircuitDesc &c, std::vector<uint32_t> &table, std::string &err) { return build_slot_table(c, table, err); }

namespace {
// word-major image scratch of the two-kernel (sweep) path
int32_t ensure_scratch(b3w_ctx *ctx) {
  if (ctx->d_scratch) return B3W_OK;
  ctx->scratch_cap = B3W_SWEEP_CHUNK;
  hipError_t e = hipMalloc((void **)&ctx->d_scratch, (size_t)ctx->scratch_cap * ctx->desc.lds_words * 4);
  if (e != hipSuccess) { ctx->scratch_cap = 0; return hip_fail(ctx, e, "hipMalloc(sweep scratch)"); }
  return B3W_OK;
}

void set_inputs(b3w_ctx *ctx) {
  struct Def { const char *name; uint32_t count; };
  static const Def comp[] = {{"h", 8}, {"m", 16}, {"t", 2}, {"b", 1}, {"d", 1}};
  static const Def nova[] = {{"n_blocks", 1}, {"block_count", 1}, {"h", 8}, {"chunk_idx_low", 1}, {"chunk_idx_high", 1},
                             {"leaf_depth", 1}, {"total_depth", 1}, {"depth", 1}, {"m", 16}, {"b", 1}};
  const Def *d = ctx->desc.kind == B3W_KIND_COMP ? comp : nova;
  const int nd = ctx->desc.kind == B3W_KIND_COMP ? 5 : 10;
  uint32_t off = 0;
  for (int i = 0; i < nd; i++) {
    ctx->inputs.push_back({d[i].name, d[i].count, off, fnv1a64(d[i].name)});
    off += d[i].count;
  }
}
}  // namespace

// The default launch shape for n witnesses into THIS buffer (no B3W_VARIANT, no autotune).
//   body streams (profiles/r02/batch_curve.json, sliced_scan_*.log): one body streams at 13 GB/s per wave, so small batches are SLICED over
//   64 ... 4 waves a body; large ones want few fat waves, compression above 6 144 witnesses the occupancy-limited 8-body variant, nova O2
//   above 32 768 steps the persistent grid;
//   the fill-ordered kernel (r06; compression, nova O2; 32-byte aligned bodies): PACED, it stores at the same rate into any memory — 7.0 TB/s
//   at 4 096 compression witnesses, 7.2 at 32 768 — where the body streams get 7.1-7.2 from a placed buffer and 5.5 from anybody else's.
//   From 128 witnesses on it is at least as fast as the sliced launch even on a placed buffer (profiles/r06/fill_small*.log), so:
//   compression -> fill order from 128 witnesses, except batches of more than 3 072 into a buffer the placement allocator KNOWS to be
//   mixed (b3w_place_is_mixed: body streams, +1 %); nova O2 (6.75-6.97 TB/s in fill order on any buffer; body streams 7.0 placed, 5.2-5.5
//   plain) -> fill order from 512 steps on, in a buffer known to be mixed from 768 to 2 560 (body streams beyond: +2 %).
int b3w_int_default_variant(const b3w_ctx *ctx, uint32_t n, const uint8_t *d_bodies, uint64_t pitch) {
  const bool comp = ctx->desc.kind == B3W_KIND_COMP, nova2 = ctx->desc.kind == B3W_KIND_NOVA_O2;
  const bool fillable = ctx->fill_ok && !(reinterpret_cast<uintptr_t>(d_bodies) & 31) && !(pitch & 31) && pitch < (1ull << 30) &&
                        (uint64_t)n * pitch + (1ull << 20) < (1ull << 37);
  if (fillable && comp && n >= 128 && (n <= 3072 || !b3w_place_is_mixed(d_bodies))) return B3W_VARIANT_REGIONFILL;
  if (fillable && nova2 && (b3w_place_is_mixed(d_bodies) ? n >= 768 && n <= 2560 : n >= 512)) return B3W_VARIANT_REGIONFILL;
  if (n <= 2560) return B3W_VARIANT_SLICED + (n <= (comp ? 32u : 8u) ? 64 : n <= 96 ? 32 : n <= 192 ? 16 : n <= 768 ? 8 : 4);
  if (comp) return n <= 6144 ? 0 : 8;
  if (nova2) return n <= 3072 ? 0 : n <= 32768 ? 3 : 4;     // (4: 8 bodies a wave on a persistent grid — 16 384 steps: -1 %, 32 768: equal, 65 536: +1.3 %)
  return 0;
}

extern "C" {

uint32_t b3w_abi_version(void) { return (1u << 16) | 0u; }

int32_t b3w_identify_wasm(const uint8_t *code, size_t len) {
  if (!code) return B3W_CIRCUIT_UNKNOWN;
  Sha256 s;
  s.update(code, len);
  uint8_t dg[32];
  s.final(dg);
  char hex[65];
  for (int i = 0; i < 32; i++) snprintf(hex + 2 * i, 3, "%02x", dg[i]);
  for (int c = 0; c < 4; c++) if (!strcmp(hex, WASM_SHA256[c])) return c;
  return B3W_CIRCUIT_UNKNOWN;
}

int32_t b3w_create(int32_t circuit, int32_t device, b3w_ctx **out) {
  if (!out || circuit < 0 || circuit > 3) return B3W_E_BAD_ARGUMENT;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return B3W_E_NO_DEVICE;
  b3w_ctx *ctx = new b3w_ctx;
  ctx->circuit = circuit;
  ctx->desc = CIRCUITS[circuit];
  ctx->device = device;
  set_inputs(ctx);
  std::vector<uint32_t> table;
  if (!build_slot_table(ctx->desc, table, ctx->last_error)) { delete ctx; return B3W_E_BAD_ARGUMENT; }
  // the fill-ordered variant (caller-owned buffers) keeps the table in LDS in 16 bits a slot (b3w_kernels.hip fill_entry16[_nova]): compression —
  // image words below 1024, no shifted word slots, no 256-bit slots; nova O2 — everything but the 67 IsZero inverses in the NARROW part of the
  // image (below B3W_LDS_WIDE), no shifted word slots, and the 256-bit slots exactly those 67 (their numbers go behind the inverse table in d_aux)
  std::vector<uint32_t> wide_slots(B3W_NOVA_ISZERO, 0xFFFFFFFFu);
  ctx->fill_ok = ctx->desc.kind == B3W_KIND_COMP || ctx->desc.kind == B3W_KIND_NOVA_O2;
  for (uint32_t i = 0; i < table.size(); i++) {
    const uint32_t e = table[i], src = e & 0xFFFu, sh = (e >> 12) & 31u, mode = (e >> 17) & 3u;
    if (mode != B3W_MODE_BIT && sh != 0) ctx->fill_ok = false;
    if (ctx->desc.kind == B3W_KIND_COMP) {
      if (src >= 1023 || mode == B3W_MODE_W256) ctx->fill_ok = false;
    } else if (mode == B3W_MODE_W256) {
      const uint32_t j = src >= B3W_LDS_WIDE && (src - B3W_LDS_WIDE) % 8 == 0 ? (src - B3W_LDS_WIDE) / 8 : 0xFFFFFFFFu;
      if (j >= B3W_NOVA_ISZERO || wide_slots[j] != 0xFFFFFFFFu) ctx->fill_ok = false; else wide_slots[j] = i;
    } else if (src + (mode == B3W_MODE_W64 ? 1u : 0u) >= B3W_LDS_WIDE) ctx->fill_ok = false;      // (the word behind src is read for every slot, used by two-word slots only)
  }
  if (ctx->desc.kind == B3W_KIND_NOVA_O2) {
    for (uint32_t v : wide_slots) if (v == 0xFFFFFFFFu) ctx->fill_ok = false;
    // the second launch writes the inverses' whole 128-byte LINES, wherever a body starts in a line: the three slots to either side of an
    // inverse must be things it has without the compression trace (step inputs and what nova_select makes of them), and the inverses' slot
    // numbers ascend (it tells a line it has written already by the slot before)
    for (uint32_t j = 0; ctx->fill_ok && j < B3W_NOVA_ISZERO; j++) {
      if (j > 0 && wide_slots[j] <= wide_slots[j - 1]) ctx->fill_ok = false;
      for (int d = -3; ctx->fill_ok && d <= 3; d++) {
        const int64_t k = (int64_t)wide_slots[j] + d;
        if (k < 0 || k >= (int64_t)ctx->desc.nwit) continue;
        const uint32_t e = table[k], src = e & 0xFFFu, mode = (e >> 17) & 3u;
        if (mode != B3W_MODE_W256 && !((src >= B3W_A_H && src < B3W_A_O) || src >= B3W_LDS_NV)) ctx->fill_ok = false;
      }
    }
  }
  // The fill-ordered kernel's own table, 16 bits a slot, made here: image word (10 bits) | shift of a BIT slot or kind of a word slot (5 bits:
  // 0 one word, 1 two, 2 a 256-bit slot) | word flag.  A nova image has 1 184 words — 11 bits — but its slots mention only some 820 of them:
  // the words from 1 024 on get an ALIAS in a word below 1 024 that no slot mentions (the tracer copies them there once a unit is traced;
  // what the trace kept in such a word is scratch by then), and the table names the alias.
  std::vector<uint16_t> ftab(table.size(), 0);
  std::vector<uint32_t> alias;                                              // from | to << 16
  if (ctx->fill_ok) {
    const bool nova = ctx->desc.kind != B3W_KIND_COMP;
    std::vector<uint8_t> used(4097, 0);                                       // (an entry's word: 12 bits)
    for (uint32_t i = 0; i < ctx->desc.nwit; i++) {
      const uint32_t e = table[i], src = e & 0xFFFu, mode = (e >> 17) & 3u;
      if (mode == B3W_MODE_W256) continue;
      used[src] = 1;
      if (mode == B3W_MODE_W64) used[src + 1] = 1;
    }
    // not to be written over: the ok / status words and what the storing waves report from (outputs; the nova outputs' words)
    used[B3W_LDS_OKWORD] = used[B3W_LDS_OKWORD + 1] = 1;
    for (uint32_t k = 0; k < 16; k++) used[B3W_A_O + k] = 1;
    if (nova) for (uint32_t k : {NV_N_BLOCKS, NV_BLOCK_COUNT_OUT, NV_TOTAL_DEPTH, NV_DEPTH_OUT, NV_CIL, NV_CIH, NV_LEAF_DEPTH}) used[B3W_LDS_NV + k] = 1;
    for (uint32_t w = 2048; w < 4097; w++) if (used[w]) ctx->fill_ok = false;     // (no alias for those: 11 bits of from)
    std::vector<uint16_t> to(2048, 0xFFFFu);
    uint32_t hole = B3W_LDS_HG;                                             // (holes among the G-function words first: the high halves of its sums)
    for (uint32_t w = 1024; w < 2048 && ctx->fill_ok; w++) {
      if (!used[w]) continue;
      while (hole < 1024 && used[hole]) hole++;
      if (hole >= 1024 || alias.size() >= B3W_ALIAS_MAX) { ctx->fill_ok = false; break; }
      used[hole] = 1; to[w] = (uint16_t)hole;
      alias.push_back(w | hole << 16);
    }
    for (uint32_t i = 0; i < table.size() && ctx->fill_ok; i++) {
      const uint32_t e = table[i], sh = (e >> 12) & 31u, mode = (e >> 17) & 3u;
      uint32_t src = e & 0xFFFu;
      if (mode == B3W_MODE_W256) { ftab[i] = (uint16_t)(0x8000u | 2u << 10); continue; }
      if (src >= 1024) {
        if (mode == B3W_MODE_W64 || src >= 2048 || to[src] == 0xFFFFu) { if (i < ctx->desc.nwit) ctx->fill_ok = false; continue; }   // (padding entries name word 0)
        src = to[src];
      }
      if (mode == B3W_MODE_W64 && src + 1 >= 1024) { ctx->fill_ok = false; continue; }
      ftab[i] = (uint16_t)(mode == B3W_MODE_BIT ? (src | sh << 10) : (src | (mode == B3W_MODE_W64 ? 1u : 0u) << 10 | 0x8000u));
    }
  }
  DeviceGuard guard(device);              // the caller's current device (torch's, say) is put back on return
  if (guard.err != hipSuccess) { delete ctx; return B3W_E_NO_DEVICE; }
  const CircuitDesc &d = ctx->desc;
  // (the 16-bit table of the fill-ordered kernel lies behind the 32-bit one: table.size() is a multiple of 32 words)
  hipError_t e = hipMalloc((void **)&ctx->d_table_base, (table.size() + 32) * 4 + ftab.size() * 2);
  if (e == hipSuccess) e = hipMemset(ctx->d_table_base, 0, 32 * 4);
  if (e == hipSuccess) ctx->d_table = ctx->d_table_base + 32;
  if (e == hipSuccess) e = hipMemcpy(ctx->d_table, table.data(), table.size() * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(ctx->d_table + table.size(), ftab.data(), ftab.size() * 2, hipMemcpyHostToDevice);
  if (e == hipSuccess && d.kind != B3W_KIND_COMP) {
    std::vector<uint32_t> aux = build_nova_aux(d.prime);
    static_assert(B3W_AUX_WIDE_SLOTS == 16 + 8 * B3W_INV_TABLE_N, "b3w_kernels.h and b3w_internal.h disagree about the inverse table");
    aux.insert(aux.end(), wide_slots.begin(), wide_slots.end());          // (O2: where the fill-ordered path's second launch writes the inverses)
    if (ctx->fill_ok && d.kind == B3W_KIND_NOVA_O2) {
      // ... and what that launch stores: per offset of a body in a 128-byte line (ph slots), the body's slots in lines that hold an inverse
      aux.resize(B3W_AUX_LINE_LISTS + 2 * 4 * B3W_LINE_LIST_MAX, 0xFFFFFFFFu);
      for (uint32_t ph = 0; ph < 4; ph++) {
        uint32_t cnt = 0, last_line = 0xFFFFFFFFu;
        for (uint32_t j = 0; j < B3W_NOVA_ISZERO; j++) {
          const uint32_t line = (ph + wide_slots[j]) >> 2;
          if (line == last_line) continue;                                   // (the slot numbers ascend: checked above)
          last_line = line;
          for (uint32_t i = 0; i < 4; i++) {
            const uint32_t k = line * 4 + i - ph;                            // wraps in front of the body
            if (k >= d.nwit) continue;
            if (cnt == B3W_LINE_LIST_MAX) { ctx->fill_ok = false; break; }
            aux[B3W_AUX_LINE_LISTS + 2 * (ph * B3W_LINE_LIST_MAX + cnt)] = k;
            aux[B3W_AUX_LINE_LISTS + 2 * (ph * B3W_LINE_LIST_MAX + cnt) + 1] = table[k];
            cnt++;
          }
        }
        aux[B3W_AUX_LINE_COUNTS + ph] = cnt;
      }
      aux.resize(B3W_AUX_ALIAS_LIST + B3W_ALIAS_MAX, 0u);                    // ... and the aliases its tracer makes
      aux[B3W_AUX_ALIAS_COUNT] = (uint32_t)alias.size();
      for (size_t k = 0; k < alias.size(); k++) aux[B3W_AUX_ALIAS_LIST + k] = alias[k];
    }
    e = hipMalloc(&ctx->d_aux, aux.size() * 4);
    if (e == hipSuccess) e = hipMemcpy(ctx->d_aux, aux.data(), aux.size() * 4, hipMemcpyHostToDevice);
  }
  {
    std::vector<uint32_t> xt;
    if (!build_exact_table(d, xt)) { b3w_destroy(ctx); return B3W_E_BAD_ARGUMENT; }
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_exact_table, xt.size() * 4);
    if (e == hipSuccess) e = hipMemcpy(ctx->d_exact_table, xt.data(), xt.size() * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_prime, 32);
    if (e == hipSuccess) e = hipMemcpy(ctx->d_prime, d.prime, 32, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_fe_inputs, 32 * 32);
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_status2, 8);
    std::vector<uint32_t> ins;
    if (build_input_slots(d, ins)) {
      if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_in_slots, ins.size() * 4);
      if (e == hipSuccess) e = hipMemcpy(ctx->d_in_slots, ins.data(), ins.size() * 4, hipMemcpyHostToDevice);
    }
  }
  if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_rec1, d.nin * 4);
  if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_body1, (size_t)d.nwit * 32);
  if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_status1, 4);
  if (e != hipSuccess) { b3w_destroy(ctx); return B3W_E_HIP; }
  const char *v = getenv("B3W_VARIANT");
  if (v) { ctx->variant = atoi(v); ctx->variant_auto = false; }
  if (ctx->variant >= B3W_VARIANT_SWEEP && ctx->variant < B3W_VARIANT_REGIONFILL && ensure_scratch(ctx) != B3W_OK) { b3w_destroy(ctx); return B3W_E_HIP; }
  *out = ctx;
  return B3W_OK;
}

void b3w_destroy(b3w_ctx *ctx) {
  if (!ctx) return;
  B3wCaptureRelaxed relaxed;                                 // (b3w_capture.h: a release may run while somebody's capture is open)
  DeviceGuard guard(ctx->device);
  (void)b3w_ctx_trim(ctx);
  if (ctx->d_table_base) (void)hipFree(ctx->d_table_base);
  if (ctx->d_aux) (void)hipFree(ctx->d_aux);
  if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
  if (ctx->d_exact_table) (void)hipFree(ctx->d_exact_table);
  if (ctx->d_prime) (void)hipFree(ctx->d_prime);
  if (ctx->d_fe_inputs) (void)hipFree(ctx->d_fe_inputs);
  if (ctx->d_status2) (void)hipFree(ctx->d_status2);
  if (ctx->d_in_slots) (void)hipFree(ctx->d_in_slots);
  if (ctx->d_rec1) (void)hipFree(ctx->d_rec1);
  if (ctx->d_body1) (void)hipFree(ctx->d_body1);
  if (ctx->d_status1) (void)hipFree(ctx->d_status1);
  delete ctx;
}

int32_t b3w_info(const b3w_ctx *ctx, uint32_t *n32, uint8_t prime_le[32], uint32_t *witness_size,
                 uint32_t *input_size, uint32_t version[3]) {
  if (!ctx) return B3W_E_BAD_ARGUMENT;
  if (n32) *n32 = 8;
  if (prime_le) memcpy(prime_le, ctx->desc.prime, 32);
  if (witness_size) *witness_size = ctx->desc.nwit;
  if (input_size) *input_size = ctx->desc.nin;
  if (version) { version[0] = 2; version[1] = 1; version[2] = 6; }   // circom 2.1.6 (WASM getVersion & co.)
  return B3W_OK;
}

int32_t b3w_input_signal_size(const b3w_ctx *ctx, uint64_t h) {
  if (!ctx) return 0;
  for (const InputSignal &s : ctx->inputs) if (s.hash == h) return (int32_t)s.count;
  return 0;
}

uint32_t b3w_public_words(const b3w_ctx *ctx) { return ctx ? ctx->desc.npub : 0; }

int32_t b3w_last_error(const b3w_ctx *ctx, char *buf, size_t len) {
  if (!ctx || !buf || !len) return B3W_E_BAD_ARGUMENT;
  snprintf(buf, len, "%s", ctx->last_error.c_str());
  return B3W_OK;
}

int32_t b3w_write_wtns_header(const b3w_ctx *ctx, uint8_t out[76]) {
  if (!ctx || !out) return B3W_E_BAD_ARGUMENT;
  uint32_t w[19];
  memcpy(&w[0], "wtns", 4);
  w[1] = 2; w[2] = 2;                      // version, number of sections
  w[3] = 1; w[4] = 8 + 32; w[5] = 0;       // section 1 id, u64 length
  w[6] = 32;                               // n8
  memcpy(&w[7], ctx->desc.prime, 32);
  w[15] = ctx->desc.nwit;
  const uint64_t len = 32ull * ctx->desc.nwit;
  w[16] = 2; w[17] = (uint32_t)len; w[18] = (uint32_t)(len >> 32);
  memcpy(out, w, 76);
  return B3W_OK;
}

int32_t b3w_batch_run_device(b3w_ctx *ctx, const uint32_t *d_records, uint32_t n, uint8_t *d_bodies, uint64_t pitch,
                             uint32_t *d_public, int32_t *d_status, void *stream) {
  if (!ctx || !d_records || !d_bodies) return B3W_E_BAD_ARGUMENT;
  const uint64_t body = 32ull * ctx->desc.nwit;
  if (pitch == 0) pitch = body;
  if (pitch < body || (pitch & 31)) { ctx->last_error = "pitch must be >= witness_size*32 and a multiple of 32"; return B3W_E_BAD_ARGUMENT; }
  if (reinterpret_cast<uintptr_t>(d_bodies) & 15) { ctx->last_error = "d_bodies must be 16-byte aligned"; return B3W_E_BAD_ARGUMENT; }
  ON_DEVICE(ctx);
  // default launch shape (profiles/r02/batch_curve.json, sliced_scan_*.log; tools/ubench/batch_curve.py, sliced_scan.py: every
  // setting at every batch size 1 ... 65 536).  One body streams at 13 GB/s per wave, so up to 2 560 witnesses a body is SLICED
  // over 64 ... 4 waves (about 4 096 store streams in flight whatever the batch: one witness 56 -> 5 us, 512 witnesses 4.4 -> 7.7
  // M/s); large batches want few fat waves, and above 6 144 compression witnesses the occupancy-limited 8-body variant wins
  // by 4-5 %
  int variant = ctx->variant;
  // (a variant picked by the autotuner on a large batch does not apply to small ones: those follow the default policy unless B3W_VARIANT says otherwise)
  if (ctx->variant_auto || (ctx->variant_tuned && n <= 2560)) variant = b3w_int_default_variant(ctx, n, d_bodies, pitch);
  if (variant >= B3W_VARIANT_REGIONFILL && !ctx->fill_ok) { ctx->last_error = "the fill-ordered variants (200, 201) exist for the compression circuit and the nova O2 builds"; return B3W_E_BAD_ARGUMENT; }
  int rc = b3w_launch_batch(ctx->desc.kind, variant, d_records, n, d_bodies, pitch, ctx->d_table, ctx->desc.nwit,
                            d_public, d_status, ctx->d_aux, ctx->d_scratch, ctx->scratch_cap, (hipStream_t)stream);
  if (rc == 0) return B3W_OK;
  if (rc == -5) { ctx->last_error = "the sweep and fill-ordered paths need 32-byte aligned bodies and pitch < 2^30"; return B3W_E_BAD_ARGUMENT; }
  if (rc < 0) { ctx->last_error = "no kernel for this circuit kind / variant"; return B3W_E_BAD_ARGUMENT; }
  return hip_fail(ctx, (hipError_t)rc, "kernel launch");
}

int32_t b3w_batch_verify_device(b3w_ctx *ctx, const uint8_t *d_bodies, uint32_t n, uint64_t pitch, uint32_t *d_mismatch,
                                void *stream) {
  if (!ctx || !d_bodies || !d_mismatch) return B3W_E_BAD_ARGUMENT;
  if (!ctx->d_in_slots) { ctx->last_error = "this circuit's layout does not keep every input as a slot"; return B3W_E_BAD_ARGUMENT; }
  const uint64_t body = 32ull * ctx->desc.nwit;
  if (pitch == 0) pitch = body;
  if (pitch < body || (pitch & 31) || (reinterpret_cast<uintptr_t>(d_bodies) & 15)) {
    ctx->last_error = "pitch must be >= witness_size*32 and a multiple of 32, bodies 16-byte aligned";
    return B3W_E_BAD_ARGUMENT;
  }
  ON_DEVICE(ctx);
  int rc = b3w_launch_verify(ctx->desc.kind, ctx->d_in_slots, n, d_bodies, pitch, ctx->d_table, ctx->desc.nwit, d_mismatch,
                             ctx->d_aux, (hipStream_t)stream);
  return rc ? hip_fail(ctx, (hipError_t)rc, "verify launch") : B3W_OK;
}

// Pick the kernel variant for THIS output buffer: the fused kernels' store pattern is sensitive to where
// the buffer sits (5.3-6.7 TB/s), the two-kernel sweep path is not (~5.5 TB/s).  All variants are
// bit-identical, so this is purely a speed choice.  Times each candidate on the caller's buffers (which are
// overwritten with the correct witnesses), keeps the fastest in the ctx.  Allocates the sweep scratch.
int32_t b3w_batch_autotune_device(b3w_ctx *ctx, const uint32_t *d_records, uint32_t n, uint8_t *d_bodies, uint64_t pitch,
                                  uint32_t *d_public, int32_t *d_status, void *stream, int32_t *chosen_variant,
                                  float *chosen_ms) {
  if (!ctx || !d_records || !d_bodies || !n) return B3W_E_BAD_ARGUMENT;
  ON_DEVICE(ctx);
  int32_t rc;
  if (n <= 2560 && (ctx->variant_auto || ctx->variant_tuned)) {
    // small batches: the sliced launch the default policy picks is the only candidate (profiles/r02/batch_curve.json: within 2 %
    // of the best shape at every size); time it and say which it is
    float ms = 0;
    rc = B3W_OK;
    for (int w = 0; w < 2 && rc == B3W_OK; w++) rc = b3w_batch_run_device(ctx, d_records, n, d_bodies, pitch, d_public, d_status, stream);
    if (rc == B3W_OK) rc = b3w_batch_time_device(ctx, d_records, n, d_bodies, pitch, d_public, d_status, stream, 5, &ms);
    if (rc) return rc;
    if (chosen_variant) *chosen_variant = b3w_int_default_variant(ctx, n, d_bodies, pitch ? pitch : 32ull * ctx->desc.nwit);
    if (chosen_ms) *chosen_ms = ms;
    return B3W_OK;
  }
  rc = ensure_scratch(ctx);
  if (rc) return rc;
  // fused with 4 (compression) / 2 (nova) bodies per wave, also with 8 (compression: for large batches occupancy-limited,
  // variant 8; nova O2: variant 3), the two-kernel sweep, and — compression — the fill-ordered fused kernel: on a placed buffer the body
  // streams win (7.2 against 6.7 TB/s), on a caller's plain buffer the fill order does (6.4 against 5.5; profiles/r06/variant_scan_*.log)
  // (nova O2: the fill-ordered path gains 4-6 % on a plain buffer — its tracer shares a SIMD with a storing wave and the nova storers do more
  // per slot —, and large batches have the persistent grid, variant 4)
  // (the fill order paced one step lighter, variant 201, is the fastest of all where it holds — and one step from the cliff: timed here, on
  // this buffer and batch, it is taken only where it holds)
  const int candidates[6] = {0, ctx->desc.kind == B3W_KIND_COMP ? (n > 6144 ? 8 : 3) : ctx->desc.kind == B3W_KIND_NOVA_O2 ? 3 : 0, B3W_VARIANT_SWEEP,
                             ctx->fill_ok ? B3W_VARIANT_REGIONFILL : 0, ctx->desc.kind == B3W_KIND_NOVA_O2 && n >= 16384 ? 4 : 0,
                             ctx->fill_ok ? B3W_VARIANT_REGIONFILL_LIGHT : 0};
  int best = ctx->variant;
  float best_ms = 1e30f;
  const int saved = ctx->variant;
  const bool saved_auto = ctx->variant_auto;
  ctx->variant_auto = false;
  // The candidates lie within a per cent or two of each other, the edge-paced fill order falls off its cliff in a launch now and then, and
  // the first candidates of a cold device run at a lower clock than the last: TWO rounds over all candidates, a candidate's better time
  // counts; twenty launches each where a launch is short.
  float cand_ms[6] = {1e30f, 1e30f, 1e30f, 1e30f, 1e30f, 1e30f};
  for (int round = 0; round < 2; round++) {
    for (int ci = 0; ci < 6; ci++) {
      const int c = candidates[ci];
      if (ci > 0 && c == candidates[0]) continue;
      if (round > 0 && cand_ms[ci] >= 1e30f) continue;     // (refused in the first round)
      ctx->variant = c;
      float ms = 0;
      rc = B3W_OK;
      for (int w = 0; w < 2 && rc == B3W_OK; w++)
        rc = b3w_batch_run_device(ctx, d_records, n, d_bodies, pitch, d_public, d_status, stream);
      if (rc == B3W_OK) rc = b3w_batch_time_device(ctx, d_records, n, d_bodies, pitch, d_public, d_status, stream, n <= 8192 ? 20 : 5, &ms);
      if (rc == B3W_E_BAD_ARGUMENT) continue;          // this path cannot take these buffers (alignment): not a candidate
      if (rc) { ctx->variant = saved; ctx->variant_auto = saved_auto; return rc; }
      if (ms < cand_ms[ci]) cand_ms[ci] = ms;
    }
  }
  for (int ci = 0; ci < 6; ci++)
    if (cand_ms[ci] < best_ms) { best_ms = cand_ms[ci]; best = candidates[ci]; }
  if (best_ms >= 1e30f) { ctx->variant = saved; ctx->variant_auto = saved_auto; return B3W_E_BAD_ARGUMENT; }
  ctx->variant = best;
  ctx->variant_tuned = true;
  if (chosen_variant) *chosen_variant = best;
  if (chosen_ms) *chosen_ms = best_ms;
  return B3W_OK;
}

int32_t b3w_batch_time_device(b3w_ctx *ctx, const uint32_t *d_records, uint32_t n, uint8_t *d_bodies, uint64_t pitch,
                              uint32_t *d_public, int32_t *d_status, void *stream, uint32_t iters, float *avg_ms) {
  if (!ctx || !avg_ms || !iters) return B3W_E_BAD_ARGUMENT;
  ON_DEVICE(ctx);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipError_t e = hipEventCreate(&e0);
  if (e == hipSuccess) e = hipEventCreate(&e1);
  if (e == hipSuccess) e = hipEventRecord(e0, (hipStream_t)stream);
  int32_t rc = B3W_OK;
  for (uint32_t i = 0; i < iters && e == hipSuccess && rc == B3W_OK; i++)
    rc = b3w_batch_run_device(ctx, d_records, n, d_bodies, pitch, d_public, d_status, stream);
  float ms = 0;
  if (e == hipSuccess && rc == B3W_OK) e = hipEventRecord(e1, (hipStream_t)stream);
  if (e == hipSuccess && rc == B3W_OK) e = hipEventSynchronize(e1);
  if (e == hipSuccess && rc == B3W_OK) e = hipEventElapsedTime(&ms, e0, e1);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (rc) return rc;
  if (e != hipSuccess) return hip_fail(ctx, e, "timing events");
  *avg_ms = ms / iters;
  return B3W_OK;
}

// The circuit runs for a complete set of inputs: the batch kernel with n = 1 for a canonical record, else the exact kernel.
static int32_t calc_witness_run(b3w_ctx *ctx, const std::vector<uint32_t> &rec, const std::vector<uint8_t> &fe, bool canonical,
                                uint8_t *out_body) {
  const CircuitDesc &d = ctx->desc;
  char msg[200];
  ON_DEVICE(ctx);
  if (canonical) {
    // canonical u32 record: the batch kernel with n = 1
    HIP_TRY(ctx, hipMemcpy(ctx->d_rec1, rec.data(), d.nin * 4, hipMemcpyHostToDevice));
    int32_t rc = b3w_batch_run_device(ctx, ctx->d_rec1, 1, ctx->d_body1, 0, nullptr, ctx->d_status1, nullptr);
    if (rc) return rc;
    int32_t st = 0;
    HIP_TRY(ctx, hipMemcpy(&st, ctx->d_status1, 4, hipMemcpyDeviceToHost));
    if (st == 0) {
      HIP_TRY(ctx, hipMemcpy(out_body, ctx->d_body1, (size_t)d.nwit * 32, hipMemcpyDeviceToHost));
      return B3W_OK;
    }
    // rejected or outside the fast-path domain: the exact kernel decides and names the assert
  }
  // field-element inputs: the exact kernel (b3w_exact.hip), still on the device
  HIP_TRY(ctx, hipMemcpy(ctx->d_fe_inputs, fe.data(), fe.size(), hipMemcpyHostToDevice));
  int rc = b3w_launch_exact(d.kind != B3W_KIND_COMP, ctx->d_fe_inputs, ctx->d_prime, ctx->d_exact_table, d.nwit,
                            ctx->d_body1, ctx->d_status2, nullptr);
  if (rc) return hip_fail(ctx, (hipError_t)rc, "exact kernel launch");
  uint32_t st2[2] = {0, 0};
  HIP_TRY(ctx, hipMemcpy(st2, ctx->d_status2, 8, hipMemcpyDeviceToHost));
  if (st2[0] != 0) {
    const char *ref = reference_trace(ctx->circuit, st2[1]);
    ctx->last_error = ref ? std::string(ref) : std::string("Assert Failed.\n") + assert_site_text(st2[1], msg, sizeof msg);
    return B3W_E_ASSERT_FAILED;
  }
  HIP_TRY(ctx, hipMemcpy(out_body, ctx->d_body1, (size_t)d.nwit * 32, hipMemcpyDeviceToHost));
  return B3W_OK;
}

// Keys are taken in the caller's order, as the reference's loader walks Object.keys (blake3_nova_js/witness_calculator.js:136-160):
// per key the size check, then its values; the circuit runs when the last missing input has been set — BEFORE the keys behind the
// completing one are looked at.  An assert therefore wins over a fault of a later key, a fault of an earlier key over the assert.
int32_t b3w_calc_witness(b3w_ctx *ctx, const uint64_t *name_hashes, const uint32_t *counts, const uint8_t *values_le32,
                         uint32_t nkeys, uint8_t *out_body) {
  if (!ctx || !name_hashes || !counts || !values_le32 || !out_body) return B3W_E_BAD_ARGUMENT;
  const CircuitDesc &d = ctx->desc;
  std::vector<uint32_t> rec(d.nin, 0);
  std::vector<uint8_t> fe((size_t)d.nin * 32, 0);       // inputs as field elements, record order
  std::vector<uint8_t> set(d.nin, 0);
  uint32_t nset = 0;
  bool canonical = true, ran = false;
  const uint8_t *v = values_le32;
  char msg[200];
  for (uint32_t k = 0; k < nkeys; k++) {
    const InputSignal *sig = nullptr;
    for (const InputSignal &s : ctx->inputs) if (s.hash == name_hashes[k]) sig = &s;
    const uint32_t size = sig ? sig->count : 0;
    if (counts[k] < size) { ctx->last_error = std::string("Not enough values for input signal ") + (sig ? sig->name : "?") + "\n"; return B3W_E_ARRAY_ACCESS; }
    if (counts[k] > size) { ctx->last_error = std::string("Too many values for input signal ") + (sig ? sig->name : "?") + "\n"; return B3W_E_TOO_MANY_SIGNALS; }
    for (uint32_t i = 0; i < size; i++, v += 32) {
      const uint32_t idx = sig->rec_off + i;
      if (set[idx]) { ctx->last_error = "Signal already set.\n"; return B3W_E_SIGNAL_ALREADY_SET; }
      memcpy(&fe[(size_t)idx * 32], v, 32);
      memcpy(&rec[idx], v, 4);
      for (int b = 4; b < 32; b++) canonical &= (v[b] == 0);
      set[idx] = 1;
      nset++;
    }
    if (!ran && size > 0 && nset == d.nin) {
      const int32_t rc = calc_witness_run(ctx, rec, fe, canonical, out_body);
      if (rc) return rc;
      ran = true;
    }
  }
  if (nset < d.nin) {
    snprintf(msg, sizeof msg, "Not all inputs have been set. Only %u out of %u", nset, d.nin);
    ctx->last_error = msg;
    return B3W_E_NOT_ALL_INPUTS;
  }
  return B3W_OK;
}

}  // extern "C"
