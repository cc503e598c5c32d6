// b3w_capi.cpp — C-ABI of libb3wit.so (include/b3wit.h): context, slot tables, launch plumbing.
// Host code only; every witness is computed by the HIP kernels in b3w_kernels.hip.  There is no
// CPU evaluation path in this library: without a HIP device b3w_create fails with B3W_E_NO_DEVICE.
#include <hip/hip_runtime_api.h>
#include <hip/hip_ext.h>
#include <errno.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/uio.h>
#include <unistd.h>
#include <algorithm>
#include <array>
#include <chrono>
#include <atomic>
#include <map>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "../../include/b3wit.h"
#include "b3w_atoms.h"
#include "b3w_kernels.h"
#include "b3w_r1cs_host.h"

struct b3w_layout_run { char kind; uint32_t slot, atom, bit0, len; };
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

const uint64_t P_BN254[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
const uint64_t P_VESTA[4] = {0x8c46eb2100000001ull, 0x224698fc0994a8ddull, 0x0ull, 0x4000000000000000ull};

struct InputSignal { const char *name; uint32_t count; uint32_t rec_off; uint64_t hash; };

// FNV-1a 64 of the signal name (witness_calculator.js:325-337)
uint64_t fnv1a64(const char *s) {
  uint64_t h = 0xCBF29CE484222325ull;
  for (; *s; ++s) { h ^= (uint8_t)*s; h *= 0x100000001B3ull; }
  return h;
}

struct CircuitDesc {
  int kind;
  uint32_t nwit, nin, npub;
  const uint64_t *prime;
  const b3w_layout_run *runs;
  uint32_t nruns;
  uint32_t lds_words;
};

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

constexpr uint32_t B3W_INV_TABLE_N = 2048;

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

struct b3w_ctx {
  int circuit = -1;
  CircuitDesc desc{};
  int device = -1;
  int variant = 0;
  bool variant_auto = true;           // no B3W_VARIANT and no autotune yet: launch shape chosen by batch size
  bool variant_tuned = false;         // `variant` comes from b3w_batch_autotune_device: it holds for large batches only
  std::vector<InputSignal> inputs;
  uint32_t *d_table = nullptr;        // slot table; 32 pad entries in front of it (expand() indexes from slot - 3)
  uint32_t *d_table_base = nullptr;
  void *d_aux = nullptr;
  uint32_t *d_scratch = nullptr;      // TRACE images of the two-kernel path
  uint32_t *d_exact_table = nullptr;  // exact (field-element) path: slot -> atom | bit<<16
  uint32_t *d_prime = nullptr;
  uint32_t *d_fe_inputs = nullptr;
  uint32_t *d_status2 = nullptr;
  uint32_t *d_in_slots = nullptr;     // VERIFY: body slot of each record word
  uint32_t scratch_cap = 0;
  // single-witness scratch
  uint32_t *d_rec1 = nullptr;
  uint8_t *d_body1 = nullptr;
  int32_t *d_status1 = nullptr;
  float plain_ms_per_gb = 0;          // the witness kernel on a plain hipMalloc buffer, measured once (b3w_bodies_alloc's sanity check)
  struct Spare { void *ptr; uint64_t bytes; int32_t placement; };
  std::vector<Spare> ring_spares;     // ring buffers of destroyed chains, reused by the next b3w_chain_create of the same size;
                                      // one size at a time, at most RING_SPARE_CAP bytes, released by b3w_ctx_trim (b3wit.h)
  std::string last_error;
};

struct b3w_batch {
  b3w_ctx *ctx = nullptr;
  uint32_t capacity = 0, n = 0;
  uint64_t pitch = 0;
  int32_t placement = B3W_PLACEMENT_PLAIN;
  uint32_t *d_recs = nullptr;
  uint8_t *d_bodies = nullptr;
  uint32_t *d_pub = nullptr;
  int32_t *d_status = nullptr;
};

namespace {
int32_t hip_fail(b3w_ctx *ctx, hipError_t e, const char *what) {
  if (ctx) ctx->last_error = std::string(what) + ": " + hipGetErrorString(e);
  return B3W_E_HIP;
}
#define HIP_TRY(ctx, call)                                  \
  do {                                                      \
    hipError_t _e = (call);                                 \
    if (_e != hipSuccess) return hip_fail(ctx, _e, #call);  \
  } while (0)

// Every entry point that touches the device runs with ctx->device current and puts the caller's device back on the
// way out: a process may hold contexts on several GPUs (or torch may have another device selected), and a launch on
// the null stream goes to whatever device is current.
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  hipError_t err = hipSuccess;
  explicit DeviceGuard(int dev) {
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != dev) {
      err = hipSetDevice(dev);
      switched = err == hipSuccess;
    }
  }
  ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
  DeviceGuard(const DeviceGuard &) = delete;
  DeviceGuard &operator=(const DeviceGuard &) = delete;
};
#define ON_DEVICE(ctx)                         \
  DeviceGuard _dev_guard((ctx)->device);       \
  if (_dev_guard.err != hipSuccess) return hip_fail(ctx, _dev_guard.err, "hipSetDevice")

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
  DeviceGuard guard(device);              // the caller's current device (torch's, say) is put back on return
  if (guard.err != hipSuccess) { delete ctx; return B3W_E_NO_DEVICE; }
  const CircuitDesc &d = ctx->desc;
  hipError_t e = hipMalloc((void **)&ctx->d_table_base, (table.size() + 32) * 4);
  if (e == hipSuccess) e = hipMemset(ctx->d_table_base, 0, 32 * 4);
  if (e == hipSuccess) ctx->d_table = ctx->d_table_base + 32;
  if (e == hipSuccess) e = hipMemcpy(ctx->d_table, table.data(), table.size() * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess && d.kind != B3W_KIND_COMP) {
    const std::vector<uint32_t> aux = build_nova_aux(d.prime);
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
  if (ctx->variant >= B3W_VARIANT_SWEEP && ensure_scratch(ctx) != B3W_OK) { b3w_destroy(ctx); return B3W_E_HIP; }
  *out = ctx;
  return B3W_OK;
}

void b3w_destroy(b3w_ctx *ctx) {
  if (!ctx) return;
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
  // (a variant picked by the autotuner on a large batch does not apply to small ones: those are sliced unless B3W_VARIANT says otherwise)
  if (ctx->variant_auto || (ctx->variant_tuned && n <= 2560)) {
    const bool comp = ctx->desc.kind == B3W_KIND_COMP;
    if (n <= 2560) variant = B3W_VARIANT_SLICED + (n <= (comp ? 32u : 8u) ? 64 : n <= 96 ? 32 : n <= 192 ? 16 : n <= 768 ? 8 : 4);
    else if (comp) variant = n <= 6144 ? 0 : 8;
    else if (ctx->desc.kind == B3W_KIND_NOVA_O2) variant = n <= 3072 ? 0 : 3;
    else variant = 0;
  }
  int rc = b3w_launch_batch(ctx->desc.kind, variant, d_records, n, d_bodies, pitch, ctx->d_table, ctx->desc.nwit,
                            d_public, d_status, ctx->d_aux, ctx->d_scratch, ctx->scratch_cap, (hipStream_t)stream);
  if (rc == 0) return B3W_OK;
  if (rc == -5) { ctx->last_error = "the sweep path needs 32-byte aligned bodies and pitch < 2^30"; return B3W_E_BAD_ARGUMENT; }
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
    const bool comp = ctx->desc.kind == B3W_KIND_COMP;
    if (chosen_variant) *chosen_variant = B3W_VARIANT_SLICED + (n <= (comp ? 32u : 8u) ? 64 : n <= 96 ? 32 : n <= 192 ? 16 : n <= 768 ? 8 : 4);
    if (chosen_ms) *chosen_ms = ms;
    return B3W_OK;
  }
  rc = ensure_scratch(ctx);
  if (rc) return rc;
  // fused with 4 (compression) / 2 (nova) bodies per wave, also with 8 (compression: for large batches occupancy-limited,
  // variant 8; nova O2: variant 3), and the two-kernel sweep
  const int candidates[3] = {0, ctx->desc.kind == B3W_KIND_COMP ? (n > 6144 ? 8 : 3) : ctx->desc.kind == B3W_KIND_NOVA_O2 ? 3 : 0, B3W_VARIANT_SWEEP};
  int best = ctx->variant;
  float best_ms = 1e30f;
  const int saved = ctx->variant;
  const bool saved_auto = ctx->variant_auto;
  ctx->variant_auto = false;
  for (int ci = 0; ci < 3; ci++) {
    const int c = candidates[ci];
    if (ci == 1 && c == candidates[0]) continue;
    ctx->variant = c;
    float ms = 0;
    rc = B3W_OK;
    for (int w = 0; w < 2 && rc == B3W_OK; w++)
      rc = b3w_batch_run_device(ctx, d_records, n, d_bodies, pitch, d_public, d_status, stream);
    if (rc == B3W_OK) rc = b3w_batch_time_device(ctx, d_records, n, d_bodies, pitch, d_public, d_status, stream, 5, &ms);
    if (rc == B3W_E_BAD_ARGUMENT) continue;          // this path cannot take these buffers (alignment): not a candidate
    if (rc) { ctx->variant = saved; ctx->variant_auto = saved_auto; return rc; }
    if (ms < best_ms) { best_ms = ms; best = c; }
  }
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

int32_t b3w_calc_witness(b3w_ctx *ctx, const uint64_t *name_hashes, const uint32_t *counts, const uint8_t *values_le32,
                         uint32_t nkeys, uint8_t *out_body) {
  if (!ctx || !name_hashes || !counts || !values_le32 || !out_body) return B3W_E_BAD_ARGUMENT;
  const CircuitDesc &d = ctx->desc;
  std::vector<uint32_t> rec(d.nin, 0);
  std::vector<uint8_t> fe((size_t)d.nin * 32, 0);       // inputs as field elements, record order
  std::vector<uint8_t> set(d.nin, 0);
  uint32_t nset = 0;
  bool canonical = true;
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
  }
  if (nset < d.nin) {
    snprintf(msg, sizeof msg, "Not all inputs have been set. Only %u out of %u", nset, d.nin);
    ctx->last_error = msg;
    return B3W_E_NOT_ALL_INPUTS;
  }
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

}  // extern "C"

namespace {
// ms per GB of bodies of ONE real witness launch filling `d_buf` (valid synthetic records, all alike: the store pattern
// is what matters); 0 when it could not be measured.  Used to check that a buffer labelled "mixed" really is faster.
float time_witness_fill(b3w_ctx *ctx, uint8_t *d_buf, uint64_t bytes) {
  const CircuitDesc &d = ctx->desc;
  const uint64_t body = 32ull * d.nwit;
  const uint32_t n = (uint32_t)std::min<uint64_t>(bytes / body, 16384);
  if (n < 256) return 0;
  std::vector<uint32_t> recs((size_t)n * d.nin, 0);
  for (uint32_t i = 0; i < n; i++) {
    uint32_t *r = &recs[(size_t)i * d.nin];
    for (uint32_t k = 0; k < d.nin; k++) r[k] = 0x9E3779B9u * (i * d.nin + k + 1);
    if (d.kind == B3W_KIND_COMP) { r[26] = 64; r[27] = 3; }
    else { r[0] = 16; r[1] = 3; r[11] = 0; r[12] = 11; r[13] = 11; r[14] = 10; r[31] = 64; }   // a leaf step at depth 10 of 11
  }
  uint32_t *d_recs = nullptr;
  int32_t *d_st = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  float ms = 0;
  hipError_t e = hipMalloc((void **)&d_recs, recs.size() * 4);
  if (e == hipSuccess) e = hipMalloc((void **)&d_st, (size_t)n * 4);
  if (e == hipSuccess) e = hipMemcpy(d_recs, recs.data(), recs.size() * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipEventCreate(&e0);
  if (e == hipSuccess) e = hipEventCreate(&e1);
  int32_t rc = B3W_OK;
  for (int it = 0; it < 8 && e == hipSuccess && rc == B3W_OK; it++) {
    if (it == 2) e = hipEventRecord(e0, nullptr);
    if (e == hipSuccess) rc = b3w_batch_run_device(ctx, d_recs, n, d_buf, body, nullptr, d_st, nullptr);
  }
  if (e == hipSuccess && rc == B3W_OK) e = hipEventRecord(e1, nullptr);
  if (e == hipSuccess && rc == B3W_OK) e = hipEventSynchronize(e1);
  if (e == hipSuccess && rc == B3W_OK) e = hipEventElapsedTime(&ms, e0, e1);
  int32_t st0 = -1;
  if (e == hipSuccess && rc == B3W_OK) e = hipMemcpy(&st0, d_st, 4, hipMemcpyDeviceToHost);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (d_recs) (void)hipFree(d_recs);
  if (d_st) (void)hipFree(d_st);
  if (e != hipSuccess) (void)hipGetLastError();
  if (e != hipSuccess || rc != B3W_OK || st0 != 0 || ms <= 0) return 0;
  return ms / 6.0f / (float)((double)n * body / 1e9);
}
}  // namespace

extern "C" {

namespace {
std::mutex g_check_mtx;
double g_check_seconds[64];                                   // per device: time spent in the real-kernel check of "mixed" buffers
struct CheckClock {
  int dev; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  explicit CheckClock(int d) : dev(d) {}
  ~CheckClock() {
    std::lock_guard<std::mutex> g(g_check_mtx);
    if (dev >= 0 && dev < 64) g_check_seconds[dev] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
};
}  // namespace

int32_t b3w_bodies_search_stats(const b3w_ctx *ctx, double out[5]) {
  if (!ctx || !out) return B3W_E_BAD_ARGUMENT;
  b3w_place_search_stats(ctx->device, out);
  std::lock_guard<std::mutex> g(g_check_mtx);
  out[4] = ctx->device >= 0 && ctx->device < 64 ? g_check_seconds[ctx->device] : 0.0;
  return B3W_OK;
}

void b3w_bodies_search_limit(double seconds) { b3w_place_search_limit(seconds); }

int32_t b3w_bodies_search_breakdown(const b3w_ctx *ctx, double out[4]) {
  if (!ctx || !out) return B3W_E_BAD_ARGUMENT;
  b3w_place_cost_breakdown(ctx->device, out);
  return B3W_OK;
}

int32_t b3w_bodies_store_rate(b3w_ctx *ctx, void *d_bodies, uint32_t n, uint64_t pitch, int32_t shape, uint32_t iters, void *stream, double *gb_per_s) {
  if (!ctx || !d_bodies || !gb_per_s) return B3W_E_BAD_ARGUMENT;
  const uint64_t body = 32ull * ctx->desc.nwit;
  if (pitch == 0) pitch = body;
  if (pitch < body || (pitch & 15) || (reinterpret_cast<uintptr_t>(d_bodies) & 15)) return B3W_E_BAD_ARGUMENT;
  ON_DEVICE(ctx);
  const int rc = b3w_place_store_rate(static_cast<uint8_t *>(d_bodies), pitch, n, (uint32_t)body, shape, iters, (hipStream_t)stream, gb_per_s);
  if (rc == -(int)hipErrorInvalidValue) return B3W_E_BAD_ARGUMENT;
  return rc ? hip_fail(ctx, (hipError_t)-rc, "store-rate launches") : B3W_OK;
}

int32_t b3w_bodies_alloc(b3w_ctx *ctx, uint64_t bytes, void **d_ptr, int32_t *placement) {
  if (!ctx || !d_ptr || !bytes) return B3W_E_BAD_ARGUMENT;
  *d_ptr = nullptr;
  if (placement) *placement = B3W_PLACEMENT_PLAIN;
  ON_DEVICE(ctx);
  const char *env = getenv("B3W_PLACEMENT");
  const bool want_mixed = !(env && !strcmp(env, "plain")) && bytes >= (512ull << 20);
  if (want_mixed) {
    int mixed = 0;
    const int rc = b3w_place_alloc(ctx->device, bytes, 1, d_ptr, &mixed, nullptr);
    if (rc == 0) {
      // "mixed" is a claim about speed: check it with the real witness kernel against a plain hipMalloc buffer
      // (measured once per context) and take the label back when the gain is below 10 % — the buffer stays usable.
      static const bool check = !(getenv("B3W_PLACE_CHECK") && !strcmp(getenv("B3W_PLACE_CHECK"), "0"));
      if (mixed && check) {
        CheckClock clock(ctx->device);
        if (ctx->plain_ms_per_gb == 0) {
          // the slowest of three distinct hipMalloc buffers: one plain buffer in eight or so straddles a class border
          // by luck and is as fast as a placed one (profiles/r02: a `--placement plain` bench run at 0.87) — that must
          // not become the yardstick.  Two of them are alive at a time, so that the next one lies elsewhere: 16 GiB at most.
          void *prev = nullptr;
          const uint64_t pb = std::min<uint64_t>(bytes, 8ull << 30);
          for (int i = 0; i < 3; i++) {
            void *cur = nullptr;
            const auto tm0 = std::chrono::steady_clock::now();
            if (hipMalloc(&cur, pb) != hipSuccess) { (void)hipGetLastError(); break; }
            const auto tm1 = std::chrono::steady_clock::now();
            if (prev) (void)hipFree(prev);
            const auto tm2 = std::chrono::steady_clock::now();
            prev = cur;
            const float one = time_witness_fill(ctx, static_cast<uint8_t *>(cur), pb);
            ctx->plain_ms_per_gb = std::max(ctx->plain_ms_per_gb, one);
            if (getenv("B3W_PLACE_DEBUG"))
              fprintf(stderr, "b3w_bodies_alloc: yardstick %d: hipMalloc %.3f s, hipFree(previous) %.3f s, fill launches %.3f s -> %.4f ms/GB\n", i,
                      std::chrono::duration<double>(tm1 - tm0).count(), std::chrono::duration<double>(tm2 - tm1).count(),
                      std::chrono::duration<double>(std::chrono::steady_clock::now() - tm2).count(), one);
          }
          if (prev) (void)hipFree(prev);
          if (ctx->plain_ms_per_gb == 0) ctx->plain_ms_per_gb = -1;          // could not measure: do not try again
        }
        if (ctx->plain_ms_per_gb > 0) {
          const float placed = time_witness_fill(ctx, static_cast<uint8_t *>(*d_ptr), std::min<uint64_t>(bytes, 8ull << 30));
          if (placed > 0 && placed > ctx->plain_ms_per_gb / 1.10f) {
            mixed = B3W_PLACEMENT_INTERLEAVED;
            if (getenv("B3W_PLACE_DEBUG"))
              fprintf(stderr, "b3w_bodies_alloc: placed buffer %.4f ms/GB against plain %.4f ms/GB: below +10 %%, reported as interleaved (no speed claim)\n", placed,
                      ctx->plain_ms_per_gb);
          } else if (getenv("B3W_PLACE_DEBUG")) {
            fprintf(stderr, "b3w_bodies_alloc: placed buffer %.4f ms/GB, plain %.4f ms/GB (%+.0f %%)\n", placed, ctx->plain_ms_per_gb,
                    placed > 0 ? (ctx->plain_ms_per_gb / placed - 1.0) * 100.0 : 0.0);
          }
        }
      }
      if (placement) *placement = mixed;                              // 0 plain, 1 mixed, 2 interleaved without the speed claim
      return B3W_OK;
    }
    (void)hipGetLastError();   // the virtual-memory path is an optimisation: fall through to a plain allocation
  }
  hipError_t e = hipMalloc(d_ptr, bytes);
  if (e != hipSuccess) { *d_ptr = nullptr; return e == hipErrorOutOfMemory ? B3W_E_NOT_ENOUGH_MEMORY : hip_fail(ctx, e, "hipMalloc(bodies)"); }
  return B3W_OK;
}

int32_t b3w_bodies_free(b3w_ctx *ctx, void *d_ptr) {
  if (!d_ptr) return B3W_OK;
  int cur = 0;
  (void)hipGetDevice(&cur);
  DeviceGuard guard(ctx ? ctx->device : cur);
  if (b3w_place_free(d_ptr) == 0) return B3W_OK;
  hipError_t e = hipFree(d_ptr);
  return e == hipSuccess ? B3W_OK : hip_fail(ctx, e, "hipFree(bodies)");
}

void b3w_bodies_trim(void) { b3w_place_trim(); }

int32_t b3w_ctx_trim(b3w_ctx *ctx) {
  if (!ctx) return B3W_E_BAD_ARGUMENT;
  DeviceGuard guard(ctx->device);
  for (const b3w_ctx::Spare &sp : ctx->ring_spares) (void)b3w_bodies_free(ctx, sp.ptr);
  ctx->ring_spares.clear();
  return B3W_OK;
}

void b3w_bodies_configure(int64_t search_gib, int64_t pool_gib) { b3w_place_configure(search_gib, pool_gib); }

int32_t b3w_bodies_stats(const b3w_ctx *ctx, uint64_t out[6]) {
  if (!ctx || !out) return B3W_E_BAD_ARGUMENT;
  b3w_place_stats(ctx->device, out);
  return B3W_OK;
}

int32_t b3w_batch_placement(const b3w_batch *b) { return b ? b->placement : B3W_PLACEMENT_PLAIN; }

int32_t b3w_batch_alloc(b3w_ctx *ctx, uint32_t capacity, uint64_t pitch, b3w_batch **out) {
  if (!ctx || !out || !capacity) return B3W_E_BAD_ARGUMENT;
  *out = nullptr;
  const uint64_t body = 32ull * ctx->desc.nwit;
  if (pitch == 0) pitch = body;
  if (pitch < body || (pitch & 31)) return B3W_E_BAD_ARGUMENT;
  b3w_batch *b = new b3w_batch;
  b->ctx = ctx; b->capacity = capacity; b->pitch = pitch;
  DeviceGuard guard(ctx->device);
  hipError_t e = guard.err;
  if (e == hipSuccess) e = hipMalloc((void **)&b->d_recs, (size_t)capacity * ctx->desc.nin * 4);
  if (e == hipSuccess) {
    const int32_t rc = b3w_bodies_alloc(ctx, (uint64_t)capacity * pitch, (void **)&b->d_bodies, &b->placement);
    if (rc != B3W_OK) { b3w_batch_free(b); return rc; }
  }
  if (e == hipSuccess) e = hipMalloc((void **)&b->d_pub, (size_t)capacity * ctx->desc.npub * 4);
  if (e == hipSuccess) e = hipMalloc((void **)&b->d_status, (size_t)capacity * 4);
  if (e != hipSuccess) { b3w_batch_free(b); return e == hipErrorOutOfMemory ? B3W_E_NOT_ENOUGH_MEMORY : hip_fail(ctx, e, "hipMalloc"); }
  *out = b;
  return B3W_OK;
}

void b3w_batch_free(b3w_batch *b) {
  if (!b) return;
  DeviceGuard guard(b->ctx->device);
  if (b->d_recs) (void)hipFree(b->d_recs);
  if (b->d_bodies) (void)b3w_bodies_free(b->ctx, b->d_bodies);
  if (b->d_pub) (void)hipFree(b->d_pub);
  if (b->d_status) (void)hipFree(b->d_status);
  delete b;
}

int32_t b3w_batch_run(b3w_batch *b, const uint32_t *host_records, uint32_t n, void *stream) {
  if (!b || !host_records || n > b->capacity) return B3W_E_BAD_ARGUMENT;
  b3w_ctx *ctx = b->ctx;
  ON_DEVICE(ctx);
  HIP_TRY(ctx, hipMemcpyAsync(b->d_recs, host_records, (size_t)n * ctx->desc.nin * 4, hipMemcpyHostToDevice, (hipStream_t)stream));
  int32_t rc = b3w_batch_run_device(ctx, b->d_recs, n, b->d_bodies, b->pitch, b->d_pub, b->d_status, stream);
  if (rc) return rc;
  HIP_TRY(ctx, hipStreamSynchronize((hipStream_t)stream));
  b->n = n;
  return B3W_OK;
}

int32_t b3w_batch_outputs(b3w_batch *b, uint32_t *host_public, int32_t *host_status) {
  if (!b || !host_public) return B3W_E_BAD_ARGUMENT;
  b3w_ctx *ctx = b->ctx;
  ON_DEVICE(ctx);
  HIP_TRY(ctx, hipMemcpy(host_public, b->d_pub, (size_t)b->n * ctx->desc.npub * 4, hipMemcpyDeviceToHost));
  if (host_status) HIP_TRY(ctx, hipMemcpy(host_status, b->d_status, (size_t)b->n * 4, hipMemcpyDeviceToHost));
  return B3W_OK;
}

int32_t b3w_batch_fetch(b3w_batch *b, uint32_t index, uint8_t *out_body) {
  if (!b || !out_body || index >= b->n) return B3W_E_BAD_ARGUMENT;
  b3w_ctx *ctx = b->ctx;
  ON_DEVICE(ctx);
  HIP_TRY(ctx, hipMemcpy(out_body, b->d_bodies + (size_t)index * b->pitch, (size_t)ctx->desc.nwit * 32, hipMemcpyDeviceToHost));
  return B3W_OK;
}

int32_t b3w_batch_verify(b3w_batch *b, uint32_t *host_mismatch) {
  if (!b || !host_mismatch) return B3W_E_BAD_ARGUMENT;
  b3w_ctx *ctx = b->ctx;
  if (!b->n) return B3W_OK;
  ON_DEVICE(ctx);
  uint32_t *d_mm = nullptr;
  HIP_TRY(ctx, hipMalloc((void **)&d_mm, (size_t)b->n * 4));
  int32_t rc = b3w_batch_verify_device(ctx, b->d_bodies, b->n, b->pitch, d_mm, nullptr);
  hipError_t e = rc == B3W_OK ? hipMemcpy(host_mismatch, d_mm, (size_t)b->n * 4, hipMemcpyDeviceToHost) : hipSuccess;
  (void)hipFree(d_mm);
  if (rc) return rc;
  return e == hipSuccess ? B3W_OK : hip_fail(ctx, e, "hipMemcpy(mismatch)");
}

int32_t b3w_batch_write_wtns(b3w_batch *b, uint32_t first, uint32_t count, const char *dir, const char *prefix,
                             uint32_t *written) {
  return b3w_batch_write_wtns_ex(b, first, count, dir, prefix, 0, written);
}

// The writer: the calling thread moves chunks of CH bodies D2H into two pinned staging buffers; `threads` writer threads take
// file numbers from one counter and write  <dir>/<prefix><index>.wtns  = 76-byte header + body with one writev each, as soon as
// the file's chunk has arrived.  D2H of chunk k + 1 runs while chunk k is being written; a staging buffer is copied into again
// when every file of its chunk has been written.
int32_t b3w_batch_write_wtns_ex(b3w_batch *b, uint32_t first, uint32_t count, const char *dir, const char *prefix, uint32_t threads,
                                uint32_t *written) {
  if (!b || !dir || !prefix || first > b->n || count > b->n - first) return B3W_E_BAD_ARGUMENT;   // (first + count wraps in u32)
  if (written) *written = 0;
  if (!count) return B3W_OK;
  b3w_ctx *ctx = b->ctx;
  const size_t body = (size_t)ctx->desc.nwit * 32;
  const uint32_t CH = 128;                                 // witnesses per staging buffer (~96 MB)
  if (!threads) {
    const char *env = getenv("B3W_WTNS_THREADS");
    threads = env ? (uint32_t)atoi(env) : 0;
    if (!threads) threads = std::min<uint32_t>(16, std::max<uint32_t>(1, std::thread::hardware_concurrency()));
  }
  threads = std::min<uint32_t>(std::min<uint32_t>(threads, 64), count);
  uint8_t hdr[76];
  b3w_write_wtns_header(ctx, hdr);
  std::vector<int32_t> st(count);
  ON_DEVICE(ctx);
  HIP_TRY(ctx, hipMemcpy(st.data(), b->d_status + first, (size_t)count * 4, hipMemcpyDeviceToHost));
  uint8_t *stage[2] = {nullptr, nullptr};
  hipStream_t cs = nullptr;
  hipEvent_t ev[2] = {nullptr, nullptr};
  auto release = [&]() {
    if (cs) (void)hipStreamSynchronize(cs);
    for (int i = 0; i < 2; i++) {
      if (stage[i]) (void)hipHostFree(stage[i]);
      if (ev[i]) (void)hipEventDestroy(ev[i]);
    }
    if (cs) (void)hipStreamDestroy(cs);
  };
  {
    hipError_t e = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
    for (int i = 0; i < 2 && e == hipSuccess; i++) {
      e = hipHostMalloc((void **)&stage[i], (size_t)std::min(CH, count) * body, hipHostMallocDefault);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming);
    }
    if (e != hipSuccess) { release(); return hip_fail(ctx, e, "staging buffers of the .wtns writer"); }
  }
  const uint32_t nchunks = (count + CH - 1) / CH;
  auto chunk_len = [&](uint32_t k) { return count - k * CH < CH ? count - k * CH : CH; };
  std::atomic<uint32_t> next{0}, ready{0}, nwritten{0};
  std::atomic<int> failed{0};
  std::vector<std::atomic<uint32_t>> done(nchunks);
  for (auto &d : done) d.store(0);
  std::mutex err_mu;
  std::string err_text;
  auto worker = [&]() {
    char path[1024];
    for (;;) {
      const uint32_t i = next.fetch_add(1);
      if (i >= count) return;
      for (uint32_t spins = 0; ready.load(std::memory_order_acquire) <= i; ++spins) {      // its chunk has not arrived yet
        if (failed.load()) return;
        if (spins > 64) std::this_thread::yield();
      }
      const uint32_t k = i / CH;
      if (st[i] == 0 && !failed.load()) {
        snprintf(path, sizeof path, "%s/%s%u.wtns", dir, prefix, first + i);
        const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0644);
        bool ok = fd >= 0;
        if (ok) {
          struct iovec iov[2] = {{hdr, 76}, {stage[k & 1] + (size_t)(i % CH) * body, body}};
          size_t left = 76 + body;
          int at = 0;
          while (ok && left) {
            const ssize_t w = writev(fd, iov + at, 2 - at);
            if (w < 0) { if (errno == EINTR) continue; ok = false; break; }
            left -= (size_t)w;
            size_t adv = (size_t)w;
            while (adv && at < 2) {
              if (adv >= iov[at].iov_len) { adv -= iov[at].iov_len; at++; }
              else { iov[at].iov_base = static_cast<uint8_t *>(iov[at].iov_base) + adv; iov[at].iov_len -= adv; adv = 0; }
            }
          }
          if (close(fd) != 0) ok = false;
        }
        if (!ok) {
          std::lock_guard<std::mutex> g(err_mu);
          if (!failed.exchange(1)) err_text = std::string("cannot write ") + path + ": " + strerror(errno);
        } else nwritten.fetch_add(1);
      }
      done[k].fetch_add(1, std::memory_order_release);
    }
  };
  std::vector<std::thread> pool;
  int32_t rc = B3W_OK;
  try {
    for (uint32_t t = 0; t < threads; t++) pool.emplace_back(worker);
  } catch (...) {
    if (pool.empty()) { release(); ctx->last_error = "cannot start a writer thread"; return B3W_E_NOT_ENOUGH_MEMORY; }
  }
  auto issue = [&](uint32_t chunk) -> hipError_t {
    hipError_t e = hipMemcpy2DAsync(stage[chunk & 1], body, b->d_bodies + (size_t)(first + chunk * CH) * b->pitch, b->pitch, body, chunk_len(chunk),
                                    hipMemcpyDeviceToHost, cs);
    if (e == hipSuccess) e = hipEventRecord(ev[chunk & 1], cs);
    return e;
  };
  uint32_t issued = 0;
  for (uint32_t k = 0; k < nchunks && rc == B3W_OK && !failed.load(); k++) {
    while (issued < nchunks && issued < k + 2 && rc == B3W_OK) {
      if (issued >= 2)                                       // its staging buffer still holds chunk issued - 2: every file written?
        for (uint32_t spins = 0; done[issued - 2].load(std::memory_order_acquire) < chunk_len(issued - 2) && !failed.load(); ++spins)
          if (spins > 64) std::this_thread::yield();
      if (failed.load()) break;
      const hipError_t e = issue(issued);
      if (e != hipSuccess) rc = hip_fail(ctx, e, "D2H");
      issued++;
    }
    if (rc != B3W_OK || failed.load()) break;
    const hipError_t e = hipEventSynchronize(ev[k & 1]);
    if (e != hipSuccess) { rc = hip_fail(ctx, e, "D2H wait"); break; }
    ready.store(k * CH + chunk_len(k), std::memory_order_release);
  }
  if (rc != B3W_OK) failed.store(1);                         // let the writers go
  for (std::thread &t : pool) t.join();
  release();
  if (failed.load() && rc == B3W_OK) { ctx->last_error = err_text; rc = B3W_E_BAD_ARGUMENT; }
  if (written) *written = nwritten.load();
  return rc;
}

void *b3w_batch_device_ptr(b3w_batch *b, uint64_t *pitch) {
  if (!b) return nullptr;
  if (pitch) *pitch = b->pitch;
  return b->d_bodies;
}

// ---------------------------------------------------------------- chained mode planner
uint64_t b3w_chain_num_chunks(uint64_t len) { return len ? (len + 1023) / 1024 : 1; }
uint64_t b3w_chain_num_leaf_steps(uint64_t len) { return len ? (len + 63) / 64 : 1; }
uint32_t b3w_chain_path_len(uint64_t chunk, uint64_t n_chunks) { return b3w_plan_path_len(chunk, n_chunks); }
uint64_t b3w_chain_num_parent_steps(uint64_t len, uint64_t first_chunk, uint64_t n_chunks_local) {
  const uint64_t n = b3w_chain_num_chunks(len);
  if (first_chunk > n || n_chunks_local > n - first_chunk) return 0;
  return b3w_plan_parent_row(first_chunk + n_chunks_local, n) - b3w_plan_parent_row(first_chunk, n);
}
uint64_t b3w_chain_parent_row(uint64_t chunk, uint64_t n_chunks) { return b3w_plan_parent_row(chunk, n_chunks); }
int32_t b3w_chain_path_provable(uint64_t chunk, uint64_t n_chunks) { return chunk < n_chunks ? b3w_plan_path_provable(chunk, n_chunks) : 0; }

int32_t b3w_chain_plan_leaves_device(b3w_ctx *ctx, const uint8_t *d_preimage, uint64_t preimage_len, uint64_t first_chunk,
                                     uint32_t n_chunks_local, uint32_t *d_records, uint32_t *d_chunk_cvs, void *stream) {
  if (!ctx || !d_preimage || !d_records || !d_chunk_cvs) return B3W_E_BAD_ARGUMENT;
  const uint64_t n = b3w_chain_num_chunks(preimage_len);
  if (first_chunk + n_chunks_local > n) { ctx->last_error = "chunk range exceeds the preimage"; return B3W_E_BAD_ARGUMENT; }
  ON_DEVICE(ctx);
  int rc = b3w_launch_plan_leaves(d_preimage, preimage_len, first_chunk, n_chunks_local, n, d_records, d_chunk_cvs, (hipStream_t)stream);
  return rc ? hip_fail(ctx, (hipError_t)rc, "plan leaves launch") : B3W_OK;
}

}  // extern "C"

namespace {
// The tree over n_chunks chunk CVs (level 0 of d_levels): levels of more than 1 024 nodes by one merge launch each, the rest —
// carries, spine and root included — by ONE launch (b3w_plan_tree_kernel; r04: one launch per level, 10 + carries for 1 MiB).
// plan_nlocal > 0: that launch also plans the parent steps of chunks [first_chunk, + plan_nlocal) into d_recs.
int32_t chain_tree(b3w_ctx *ctx, uint32_t *d_levels, uint64_t n_chunks, uint32_t *d_root, uint64_t first_chunk, uint32_t plan_nlocal,
                   uint32_t last_blocks, uint32_t *d_recs, hipStream_t st) {
  if (n_chunks == 1) {
    HIP_TRY(ctx, hipMemcpyAsync(d_root, d_levels, 32, hipMemcpyDeviceToDevice, st));
    return B3W_OK;
  }
  const uint32_t l0 = b3w_plan_tree_first_level(n_chunks);
  uint32_t *level = d_levels;
  uint64_t count = n_chunks;
  for (uint32_t l = 0; l < l0; l++) {                          // an odd node out stays where it is: a carry the tree kernel picks up
    int rc = b3w_launch_plan_merge(level, level + 8, 16, count / 2, 0u, level + count * 8, st);
    if (rc) return hip_fail(ctx, (hipError_t)rc, "plan merge launch");
    level += count * 8;
    count /= 2;
  }
  int rc = b3w_launch_plan_tree(d_levels, n_chunks, l0, d_root, first_chunk, plan_nlocal, last_blocks, d_recs, st);
  return rc ? hip_fail(ctx, (hipError_t)rc, "plan tree launch") : B3W_OK;
}
}  // namespace

extern "C" {

int32_t b3w_chain_tree_device(b3w_ctx *ctx, uint32_t *d_levels, uint64_t n_chunks, uint32_t *d_root, void *stream) {
  if (!ctx || !d_levels || !d_root || !n_chunks) return B3W_E_BAD_ARGUMENT;
  ON_DEVICE(ctx);
  return chain_tree(ctx, d_levels, n_chunks, d_root, 0, 0, 0, nullptr, (hipStream_t)stream);
}

int32_t b3w_chain_plan_parents_device(b3w_ctx *ctx, const uint32_t *d_levels, uint64_t n_chunks, uint64_t preimage_len,
                                      uint64_t first_chunk, uint32_t n_chunks_local, uint32_t *d_records, void *stream) {
  if (!ctx || !d_levels || !d_records) return B3W_E_BAD_ARGUMENT;
  if (n_chunks != b3w_chain_num_chunks(preimage_len) || first_chunk + n_chunks_local > n_chunks) {
    ctx->last_error = "n_chunks must match the preimage and the chunk range lie inside it";
    return B3W_E_BAD_ARGUMENT;
  }
  ON_DEVICE(ctx);
  const uint64_t last_bytes = preimage_len > (n_chunks - 1) * 1024 ? preimage_len - (n_chunks - 1) * 1024 : 0;
  const uint32_t last_blocks = last_bytes ? (uint32_t)((last_bytes + 63) / 64) : 1;
  int rc = b3w_launch_plan_parents(d_levels, n_chunks, first_chunk, n_chunks_local, last_blocks, d_records, (hipStream_t)stream);
  return rc ? hip_fail(ctx, (hipError_t)rc, "plan parents launch") : B3W_OK;
}

// ---------------------------------------------------------------- rank-1 constraint check (on-device consumer #1)
}  // extern "C"

constexpr size_t R1CS_SCRATCH_STREAMS = 8;           // deferred-row scratches (27 MB each) one constraint system keeps, one per stream

struct b3w_r1cs {
  b3w_ctx *ctx = nullptr;
  uint32_t m = 0, nwires = 0, npubout = 0, npubin = 0, nprvin = 0;
  uint64_t nterms = 0;
  B3wField field{};
  uint32_t *d_rows = nullptr, *d_row_id = nullptr, *d_wires = nullptr, *d_coefR = nullptr;
  uint16_t *d_cids = nullptr;
  // tile formulation (b3w_r1cs.hip): used when every tile of B3W_R1CS_TILE wires needs at most that many outside wires
  bool tiled = false;
  uint32_t ntiles = 0, max_ext = 0, max_tile_terms = 0, ncoef = 0;
  uint32_t *d_tiles = nullptr, *d_ext = nullptr, *d_trows = nullptr, *d_trow_id = nullptr, *d_terms = nullptr, *d_tile_terms = nullptr;
  long long *d_coef_small = nullptr;
  // the lean kernel pair: the system as the kernels take it, and the deferred-row scratch — one per stream a check was
  // enqueued on, allocated at the first check on that stream and kept (a fixed size: checks go in slabs of B3W_R1CS_SLAB)
  uint32_t max_tile_rows = 0;
  uint32_t *d_trow_k = nullptr, *d_lrows = nullptr, *d_lterms = nullptr, *d_ltile_terms = nullptr;   // (its own rows and term stream: bit runs folded)
  uint32_t *d_srows = nullptr, *d_sgdesc = nullptr, *d_sgwords = nullptr, *d_sgmeta = nullptr;       // the stream kernel's program
  unsigned long long *d_smask = nullptr, *d_scost = nullptr;
  B3wR1csSystem sys{};
  // the walk kernel's program, and the system as the deferred kernel sees it behind the walk kernel (walk row order)
  bool has_walk = false;
  uint32_t *d_wtile = nullptr, *d_wruns = nullptr, *d_wrun_row = nullptr, *d_went_w = nullptr, *d_went_m = nullptr, *d_wrow_k = nullptr, *d_wrow_id = nullptr,
           *d_wtiles4 = nullptr, *d_wstatic_k = nullptr, *d_wstatic_id = nullptr;
  uint16_t *d_wexp = nullptr;
  unsigned long long *d_wmask = nullptr, *d_wstatic = nullptr;
  B3wWalk walk{};
  B3wR1csSystem sysw{};
  // ... at most R1CS_SCRATCH_STREAMS of them: the least recently used one goes when another stream comes (after the event that
  // follows its last check; a scratch a stream capture has seen stays, its graph may be replayed any time)
  struct Scratch { void *stream; unsigned long long *buf; hipEvent_t done; uint64_t tick; bool pinned; };
  mutable std::mutex scratch_mu;
  mutable std::vector<Scratch> scratch;
  mutable uint64_t scratch_tick = 0;
};


extern "C" {

static int32_t r1cs_create_impl(b3w_ctx *ctx, const uint8_t *img, size_t len, b3w_r1cs **out);

int32_t b3w_r1cs_create(b3w_ctx *ctx, const uint8_t *img, size_t len, b3w_r1cs **out) {
  if (!ctx || !img || !out) return B3W_E_BAD_ARGUMENT;
  *out = nullptr;
  try {                                                   // the image is untrusted input: no exception may cross the C boundary
    return r1cs_create_impl(ctx, img, len, out);
  } catch (const std::bad_alloc &) {
    ctx->last_error = "r1cs: not enough host memory for this image";
    return B3W_E_NOT_ENOUGH_MEMORY;
  } catch (...) {
    ctx->last_error = "r1cs: malformed image";
    return B3W_E_BAD_ARGUMENT;
  }
}

static int32_t r1cs_create_impl(b3w_ctx *ctx, const uint8_t *img, size_t len, b3w_r1cs **out) {
  B3wR1csHost H;                                          // (parsing and tiling: b3w_r1cs_host.cpp, no device involved)
  if (!b3w_r1cs_host_build(img, len, reinterpret_cast<const uint8_t *>(ctx->desc.prime), ctx->desc.nwit, &H)) { ctx->last_error = H.error; return B3W_E_BAD_ARGUMENT; }
  b3w_r1cs *r = new b3w_r1cs;
  r->ctx = ctx; r->m = H.m; r->nwires = H.nwires; r->npubout = H.npubout; r->npubin = H.npubin; r->nprvin = H.nprvin; r->nterms = H.nterms;
  r->field = H.field;
  DeviceGuard guard(ctx->device);
  hipError_t e = guard.err;
  auto up = [&](void **d, const void *src, size_t bytes) {
    if (e == hipSuccess) e = hipMalloc(d, bytes ? bytes : 4);
    if (e == hipSuccess && bytes) e = hipMemcpy(*d, src, bytes, hipMemcpyHostToDevice);
  };
  up((void **)&r->d_rows, H.rowdesc.data(), H.rowdesc.size() * 4);
  up((void **)&r->d_row_id, H.row_id.data(), H.row_id.size() * 4);
  up((void **)&r->d_wires, H.wires.data(), H.wires.size() * 4);
  up((void **)&r->d_cids, H.cids.data(), H.cids.size() * 2);
  up((void **)&r->d_coefR, H.coefR.data(), H.coefR.size() * 4);
  r->tiled = H.tiled; r->ntiles = H.ntiles; r->max_ext = H.max_ext;
  r->max_tile_terms = H.max_tile_terms;
  r->ncoef = H.ncoef;
  if (H.tiled) {
    up((void **)&r->d_tiles, H.tdesc.data(), H.tdesc.size() * 4);
    up((void **)&r->d_tile_terms, H.ttdesc.data(), H.ttdesc.size() * 4);
    up((void **)&r->d_ext, H.text.data(), H.text.size() * 4);
    up((void **)&r->d_trows, H.trows.data(), H.trows.size() * 4);
    up((void **)&r->d_trow_id, H.trow_id.data(), H.trow_id.size() * 4);
    up((void **)&r->d_terms, H.tterms.data(), H.tterms.size() * 4);
    up((void **)&r->d_coef_small, H.coef_small.data(), H.coef_small.size() * 8);
    up((void **)&r->d_trow_k, H.trow_k.data(), H.trow_k.size() * 4);
    up((void **)&r->d_lrows, H.lrows.data(), H.lrows.size() * 4);
    up((void **)&r->d_lterms, H.lterms.data(), H.lterms.size() * 4);
    up((void **)&r->d_ltile_terms, H.ltdesc.data(), H.ltdesc.size() * 4);
    r->max_tile_rows = H.max_tile_rows;
    up((void **)&r->d_srows, H.srows.data(), H.srows.size() * 4);
    up((void **)&r->d_sgdesc, H.sgdesc.data(), H.sgdesc.size() * 4);
    up((void **)&r->d_sgwords, H.sgwords.data(), H.sgwords.size() * 4);
    up((void **)&r->d_sgmeta, H.sgmeta.data(), H.sgmeta.size() * 4);
    up((void **)&r->d_smask, H.smask.data(), H.smask.size() * 8);
    up((void **)&r->d_scost, H.scost.data(), H.scost.size() * 8);
    r->sys = B3wR1csSystem{H.nwires, H.ntiles, H.max_ext, H.max_lean_terms, H.max_tile_rows, r->ncoef, r->d_tiles, r->d_ltile_terms, r->d_ext, r->d_lrows,
                           r->d_trow_id, r->d_trow_k, r->d_lterms, r->d_coefR, r->d_coef_small, r->d_rows, r->d_wires, r->d_cids,
                           H.max_g_words, H.max_g_rows, H.smask_groups, r->d_srows, r->d_sgdesc, r->d_sgwords, r->d_sgmeta, r->d_smask, r->d_scost};
  }
  if (H.tiled && H.walk) {
    up((void **)&r->d_wtile, H.wtile.data(), H.wtile.size() * 4);
    up((void **)&r->d_wmask, H.wmask.data(), H.wmask.size() * 8);
    up((void **)&r->d_wexp, H.wexp.data(), H.wexp.size() * 2);
    up((void **)&r->d_wruns, H.wruns.data(), H.wruns.size() * 4);
    up((void **)&r->d_wrun_row, H.wrun_row.data(), H.wrun_row.size() * 4);
    up((void **)&r->d_went_w, H.went_w.data(), H.went_w.size() * 4);
    up((void **)&r->d_went_m, H.went_m.data(), H.went_m.size() * 4);
    up((void **)&r->d_wrow_k, H.wrow_k.data(), H.wrow_k.size() * 4);
    up((void **)&r->d_wrow_id, H.wrow_id.data(), H.wrow_id.size() * 4);
    up((void **)&r->d_wtiles4, H.wtiles4.data(), H.wtiles4.size() * 4);
    up((void **)&r->d_wstatic, H.wstatic.data(), H.wstatic.size() * 8);
    const std::vector<uint32_t> &sk = H.wstatic_list, &sid = H.wstatic_ids;      // (b3w_r1cs_host.h: the always-deferred rows as the deferred kernel walks them)
    up((void **)&r->d_wstatic_k, sk.data(), sk.size() * 4);
    up((void **)&r->d_wstatic_id, sid.data(), sid.size() * 4);
    // SIGNED elements (b3w_r1cs_walk.hip, walk_pack): the instantiation that takes an element p - k for the small number -k.  Which
    // bodies hold such elements is a property of the circuit BUILD the context stands for (a system over this context has its witness
    // size, so it is checked against this build's bodies, whoever derived it): the circomkit nova build keeps its differences as wires
    // (121 rows a step over such values); the O2 builds and blake3_compression hold none, and for them the signed instantiation is 1 - 2 %
    // slower (profiles/r04/walk_ab_signed_elements.log, ab_signed_compression_rocprof.log).  B3W_R1CS_SIGNED=0/1 overrides;
    // H.wlinear_rows (linear rows: no optimiser has been over the system) is the static hint a foreign build would be judged by.
    const char *sg_env = getenv("B3W_R1CS_SIGNED");
    const uint32_t signed_elems = sg_env ? (atoi(sg_env) ? 1u : 0u) : (ctx->desc.kind == B3W_KIND_NOVA_O1 ? 1u : 0u);
    r->walk = B3wWalk{H.wunits, H.wexp_slots, H.wmax_gen, H.wmax_ent, r->ncoef, H.wstatic_words, H.wmax_rows, signed_elems, r->d_wtile, r->d_wmask, r->d_wexp,
                      reinterpret_cast<const uint4 *>(r->d_wruns), r->d_wrun_row, r->d_went_w, r->d_went_m, r->d_wrow_id, r->d_wstatic, r->d_coef_small,
                      r->d_wstatic_k, r->d_wstatic_id, (uint32_t)sid.size(), 0u, {sk.empty() ? 0u : sk[0], sk.empty() ? 0u : sk[1], sk.empty() ? 0u : sk[2], sk.empty() ? 0u : sk[3]}, {}};
    memcpy(r->walk.p, H.field.p, 32);
    r->sysw = r->sys;
    r->sysw.tiles = r->d_wtiles4; r->sysw.row_k = r->d_wrow_k; r->sysw.row_id = r->d_wrow_id; r->sysw.max_tile_rows = H.wmax_rows;
    r->sysw.ntiles = H.wunits;                               // (the deferred kernel's blocks are per unit)
    r->has_walk = true;
  }
  if (e != hipSuccess) { b3w_r1cs_destroy(r); return e == hipErrorOutOfMemory ? B3W_E_NOT_ENOUGH_MEMORY : hip_fail(ctx, e, "r1cs upload"); }
  *out = r;
  return B3W_OK;
}

int32_t b3w_r1cs_info(const b3w_r1cs *r, uint32_t *n_constraints, uint32_t *n_wires, uint64_t *n_terms, uint32_t *n_pub_out,
                      uint32_t *n_pub_in, uint32_t *n_prv_in) {
  if (!r) return B3W_E_BAD_ARGUMENT;
  if (n_constraints) *n_constraints = r->m;
  if (n_wires) *n_wires = r->nwires;
  if (n_terms) *n_terms = r->nterms;
  if (n_pub_out) *n_pub_out = r->npubout;
  if (n_pub_in) *n_pub_in = r->npubin;
  if (n_prv_in) *n_prv_in = r->nprvin;
  return B3W_OK;
}

int32_t b3w_r1cs_is_tiled(const b3w_r1cs *r) { return r && r->tiled ? 1 : 0; }

void b3w_r1cs_destroy(b3w_r1cs *r) {
  if (!r) return;
  DeviceGuard guard(r->ctx->device);
  if (r->d_rows) (void)hipFree(r->d_rows);
  if (r->d_row_id) (void)hipFree(r->d_row_id);
  if (r->d_wires) (void)hipFree(r->d_wires);
  if (r->d_cids) (void)hipFree(r->d_cids);
  if (r->d_coefR) (void)hipFree(r->d_coefR);
  for (uint32_t *q : {r->d_tiles, r->d_ext, r->d_trows, r->d_trow_id, r->d_terms, r->d_tile_terms}) if (q) (void)hipFree(q);
  if (r->d_coef_small) (void)hipFree(r->d_coef_small);
  for (uint32_t *q : {r->d_trow_k, r->d_lrows, r->d_lterms, r->d_ltile_terms, r->d_srows, r->d_sgdesc, r->d_sgwords, r->d_sgmeta}) if (q) (void)hipFree(q);
  if (r->d_smask) (void)hipFree(r->d_smask);
  if (r->d_scost) (void)hipFree(r->d_scost);
  for (uint32_t *q : {r->d_wtile, r->d_wruns, r->d_wrun_row, r->d_went_w, r->d_went_m, r->d_wrow_k, r->d_wrow_id, r->d_wtiles4, r->d_wstatic_k, r->d_wstatic_id}) if (q) (void)hipFree(q);
  if (r->d_wexp) (void)hipFree(r->d_wexp);
  if (r->d_wmask) (void)hipFree(r->d_wmask);
  if (r->d_wstatic) (void)hipFree(r->d_wstatic);
  for (auto &sc : r->scratch) {
    if (sc.done) { (void)hipEventSynchronize(sc.done); (void)hipEventDestroy(sc.done); }
    if (sc.buf) (void)hipFree(sc.buf);
  }
  delete r;
}

int32_t b3w_r1cs_check_device(b3w_ctx *ctx, const b3w_r1cs *r, const uint8_t *d_bodies, uint32_t n, uint64_t pitch,
                              uint32_t *d_violations, uint32_t *d_first, void *stream) {
  if (!ctx || !r || r->ctx != ctx || !d_bodies || !d_violations) return B3W_E_BAD_ARGUMENT;
  const uint64_t body = 32ull * ctx->desc.nwit;
  if (pitch == 0) pitch = body;
  if (pitch < body || (pitch & 15) || (reinterpret_cast<uintptr_t>(d_bodies) & 15)) {
    ctx->last_error = "pitch must be >= witness_size*32 and a multiple of 16, bodies 16-byte aligned";
    return B3W_E_BAD_ARGUMENT;
  }
  ON_DEVICE(ctx);
  // the walk kernel where the system fits it, else the stream kernel, else the lean pair, else the gather kernel.
  // B3W_R1CS_GATHER picks another formulation, for comparison: 1 the gather kernel, 3 the lean pair, 4 the stream kernel (round 3's default)
  static const int other = getenv("B3W_R1CS_GATHER") ? atoi(getenv("B3W_R1CS_GATHER")) : 0;
  if (r->tiled && (other == 0 || other == 3 || other == 4)) {
    unsigned long long *scratch = nullptr;
    hipEvent_t done = nullptr;
    {
      std::lock_guard<std::mutex> lock(r->scratch_mu);
      hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
      if (stream) (void)hipStreamIsCapturing((hipStream_t)stream, &cap);          // (the null stream cannot be captured)
      (void)hipGetLastError();
      const bool capturing = cap != hipStreamCaptureStatusNone;
      b3w_r1cs::Scratch *hit = nullptr;
      for (auto &sc : r->scratch) if (sc.stream == stream) hit = &sc;
      if (!hit) {
        // the first check on a stream allocates that stream's deferred-row scratch — not something a capture may contain
        if (capturing) { ctx->last_error = "the first constraint check on a stream allocates its scratch: run one check on this stream before capturing it"; return B3W_E_BAD_ARGUMENT; }
        if (r->scratch.size() >= R1CS_SCRATCH_STREAMS) {                           // a caller cycling through streams: the least recently used goes
          size_t lru = r->scratch.size();
          for (size_t k = 0; k < r->scratch.size(); k++)
            if (!r->scratch[k].pinned && (lru == r->scratch.size() || r->scratch[k].tick < r->scratch[lru].tick)) lru = k;
          if (lru == r->scratch.size()) { ctx->last_error = "every scratch of this constraint system belongs to a captured stream"; return B3W_E_NOT_ENOUGH_MEMORY; }
          (void)hipEventSynchronize(r->scratch[lru].done);                         // its last check has finished (its stream may be gone by now)
          (void)hipEventDestroy(r->scratch[lru].done);
          (void)hipFree(r->scratch[lru].buf);
          r->scratch.erase(r->scratch.begin() + lru);
        }
        b3w_r1cs::Scratch sc{stream, nullptr, nullptr, 0, false};
        HIP_TRY(ctx, hipMalloc((void **)&sc.buf, std::max(b3w_r1cs_scratch_bytes(&r->sys), r->has_walk ? b3w_r1cs_walk_scratch_bytes(&r->walk) : (size_t)0)));
        hipError_t ee = hipEventCreateWithFlags(&sc.done, hipEventDisableTiming);
        if (ee != hipSuccess) { (void)hipFree(sc.buf); return hip_fail(ctx, ee, "hipEventCreate(r1cs scratch)"); }
        r->scratch.push_back(sc);
        hit = &r->scratch.back();
      }
      hit->tick = ++r->scratch_tick;
      if (capturing) hit->pinned = true;
      scratch = hit->buf;
      done = capturing ? nullptr : hit->done;                                      // (a captured check stays made of kernel nodes only)
    }
    int lrc = -6;
    if (other == 0 && r->has_walk) lrc = b3w_launch_r1cs_walk(d_bodies, n, pitch, &r->walk, &r->sysw, &r->field, scratch, d_violations, d_first, (hipStream_t)stream);
    if (lrc == -6 && other != 3) lrc = b3w_launch_r1cs_stream(d_bodies, n, pitch, &r->sys, &r->field, scratch, d_violations, d_first, (hipStream_t)stream);
    if (lrc == -6) lrc = b3w_launch_r1cs_lean(d_bodies, n, pitch, &r->sys, &r->field, scratch, d_violations, d_first, (hipStream_t)stream);
    if (done) (void)hipEventRecord(done, (hipStream_t)stream);
    return lrc ? hip_fail(ctx, (hipError_t)lrc, "r1cs check launch") : B3W_OK;
  }
  const int rc = b3w_launch_r1cs(d_bodies, n, pitch, r->m, r->d_rows, r->d_row_id, r->d_wires, r->d_cids, r->d_coefR, &r->field, d_violations, d_first,
                                 (hipStream_t)stream);
  return rc ? hip_fail(ctx, (hipError_t)rc, "r1cs check launch") : B3W_OK;
}

int32_t b3w_batch_r1cs_check(b3w_batch *b, const b3w_r1cs *r, uint32_t *host_violations, uint32_t *host_first) {
  if (!b || !r || !host_violations) return B3W_E_BAD_ARGUMENT;
  b3w_ctx *ctx = b->ctx;
  if (!b->n) return B3W_OK;
  ON_DEVICE(ctx);
  uint32_t *d = nullptr;
  HIP_TRY(ctx, hipMalloc((void **)&d, (size_t)b->n * 8));
  int32_t rc = b3w_r1cs_check_device(ctx, r, b->d_bodies, b->n, b->pitch, d, d + b->n, nullptr);
  hipError_t e = rc == B3W_OK ? hipMemcpy(host_violations, d, (size_t)b->n * 4, hipMemcpyDeviceToHost) : hipSuccess;
  if (rc == B3W_OK && e == hipSuccess && host_first) e = hipMemcpy(host_first, d + b->n, (size_t)b->n * 4, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (rc) return rc;
  return e == hipSuccess ? B3W_OK : hip_fail(ctx, e, "hipMemcpy(r1cs check)");
}

void b3w_r1cs_consumer(void *user, const uint8_t *d_bodies, uint64_t pitch, uint64_t first_step, uint32_t count, void *stream) {
  b3w_r1cs_sink *sink = static_cast<b3w_r1cs_sink *>(user);
  if (!sink || !sink->ctx || !sink->r1cs || !sink->d_violations) return;
  const int32_t rc = b3w_r1cs_check_device(sink->ctx, sink->r1cs, d_bodies, count, pitch, sink->d_violations + first_step, nullptr, stream);
  if (rc && !sink->error) sink->error = rc;
  if (sink->next) sink->next(sink->next_user, d_bodies, pitch, first_step, count, stream);
}

// ---------------------------------------------------------------- commitments (on-device consumer #2)
}  // extern "C"

struct b3w_commit_key {
  b3w_ctx *ctx = nullptr;
  B3wCurve curve{};
  uint32_t first_slot = 0, nwin = 0, window = 0;
  uint32_t nslots = 0;                                // committed slots: first_slot .. witness_size - 1
  uint32_t *d_slotdesc = nullptr;                     // per committed slot: first virtual slot (= its bit in the packed witness) | width code << 24
  uint32_t *d_runs = nullptr; uint32_t nruns = 0;     // the same bit string as pieces of TRACE-image words (records mode)
  uint32_t *d_images = nullptr; uint32_t images_cap = 0;   // records mode: TRACE images of one chunk, word-major, grown on demand
  uint32_t *d_table = nullptr;                        // per window of `window` virtual slots: 2^window - 1 subset sums
  uint32_t *d_invtab = nullptr; uint32_t inv_nk = 0;  // O2 nova circuits: per IsZero gadget the points of +-1/k for |k| <= inv_nk
  uint32_t *d_invmeta = nullptr;                      // ... and what bodies mode needs to use them: [0, 67) the committed slot of gadget j's inverse,
                                                      // [67, 71) the witness slots of n_blocks, block_count, total_depth, depth, [71, 138) the slot's first virtual slot
  uint32_t *d_sums = nullptr;                         // Jacobian sums between the two kernels, grown on demand
  uint32_t sums_cap = 0;
  unsigned long long *d_counts = nullptr;             // b3w_commit_key_count: mixed additions of the launches made while counting
  uint64_t host_witnesses = 0;                        // ... and the witnesses of those launches
  bool counting = false;
};

namespace {
// 256-bit helpers for the curve constants (host, little-endian u32 limbs)
bool u256_geq(const uint32_t a[8], const uint32_t b[8]) { for (int i = 7; i >= 0; --i) if (a[i] != b[i]) return a[i] > b[i]; return true; }
void u256_sub_host(uint32_t a[8], const uint32_t b[8]) {
  uint64_t br = 0;
  for (int i = 0; i < 8; i++) { const uint64_t t = (uint64_t)a[i] - b[i] - br; a[i] = (uint32_t)t; br = (t >> 63) & 1; }
}
void u256_double_mod(uint32_t a[8], const uint32_t p[8]) {            // a = 2a mod p (a < p < 2^255)
  uint32_t c = 0;
  for (int i = 0; i < 8; i++) { const uint32_t n = (a[i] << 1) | c; c = a[i] >> 31; a[i] = n; }
  if (c || u256_geq(a, p)) u256_sub_host(a, p);
}
B3wCurve make_curve(const uint64_t p64[4]) {
  B3wCurve C{};
  memcpy(C.p, p64, 32);
  uint32_t x[8] = {1, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 256; i++) u256_double_mod(x, C.p);
  memcpy(C.one, x, 32);
  for (int i = 0; i < 256; i++) u256_double_mod(x, C.p);
  memcpy(C.r2, x, 32);
  memcpy(C.pm2, C.p, 32);
  const uint32_t two[8] = {2, 0, 0, 0, 0, 0, 0, 0};
  u256_sub_host(C.pm2, two);
  uint32_t inv = C.p[0];                                             // Newton: inv = p^-1 mod 2^32
  for (int i = 0; i < 5; i++) inv *= 2u - C.p[0] * inv;
  C.inv = 0u - inv;
  return C;
}
const uint64_t Q_BN254[4] = {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
const uint64_t P_VESTA_BASE[4] = {0x992d30ed00000001ull, 0x224698fc094cf91bull, 0x0ull, 0x4000000000000000ull};
}  // namespace

extern "C" {

int32_t b3w_commit_key_create(b3w_ctx *ctx, int32_t curve, uint32_t first_slot, const uint8_t *host_generators, b3w_commit_key **out) {
  return b3w_commit_key_create_ex(ctx, curve, first_slot, host_generators, 0, out);
}

uint32_t b3w_commit_key_window(const b3w_commit_key *key) { return key ? key->window : 0; }

int32_t b3w_commit_key_create_ex(b3w_ctx *ctx, int32_t curve, uint32_t first_slot, const uint8_t *host_generators, uint32_t window_bits,
                                 b3w_commit_key **out) {
  return b3w_commit_key_create_folded(ctx, curve, first_slot, host_generators, nullptr, window_bits, out);
}

int32_t b3w_slot_widths(b3w_ctx *ctx, uint16_t *out_bits) {
  if (!ctx || !out_bits) return B3W_E_BAD_ARGUMENT;
  std::vector<uint32_t> table;
  if (!build_slot_table(ctx->desc, table, ctx->last_error)) return B3W_E_BAD_ARGUMENT;
  for (uint32_t i = 0; i < ctx->desc.nwit; i++) {
    const uint32_t mode = (table[i] >> 17) & 3u;
    out_bits[i] = mode == B3W_MODE_BIT ? 1 : mode == B3W_MODE_W32 ? 32 : mode == B3W_MODE_W64 ? 64 : 256;
  }
  return B3W_OK;
}

int32_t b3w_commit_key_create_folded(b3w_ctx *ctx, int32_t curve, uint32_t first_slot, const uint8_t *host_generators,
                                     const uint8_t *folded /* per committed slot, or null */, uint32_t window_bits, b3w_commit_key **out) {
  if (!ctx || !out || !host_generators || (curve != B3W_CURVE_BN254_G1 && curve != B3W_CURVE_VESTA) || first_slot >= ctx->desc.nwit ||
      (window_bits != 0 && window_bits != B3W_COMMIT_WINDOW_SMALL && window_bits != B3W_COMMIT_WINDOW_LARGE)) {
    if (ctx) ctx->last_error = "commit key: curve 0/1, first_slot < witness_size, window_bits 0 (auto), 12 or 16";
    return B3W_E_BAD_ARGUMENT;
  }
  *out = nullptr;
  std::vector<uint32_t> table;
  if (!build_slot_table(ctx->desc, table, ctx->last_error)) return B3W_E_BAD_ARGUMENT;
  const uint32_t nslots = ctx->desc.nwit - first_slot;
  // virtual slots: one per bit a slot can hold (BIT 1, W32 32, W64 64, W256 256)
  std::vector<uint32_t> nbits(nslots), first_v(nslots);
  uint64_t nv = 0;                                     // virtual slots = bits of the packed witness
  for (uint32_t i = 0; i < nslots; i++) {
    const uint32_t mode = (table[first_slot + i] >> 17) & 3u;
    nbits[i] = mode == B3W_MODE_BIT ? 1u : mode == B3W_MODE_W32 ? 32u : mode == B3W_MODE_W64 ? 64u : 256u;
    if (folded && folded[i] == 1) nbits[i] = 0;       // folded into other slots' generators by the caller: no virtual slots, no points
    else if (folded && (folded[i] & 0x80)) {          // only bit (folded[i] & 31) of this 32-bit word is committed, with the generator given
      if (mode != B3W_MODE_W32 || (folded[i] & 0x60)) { ctx->last_error = "commit key: a single-bit fold needs a 32-bit slot and a bit below 32"; return B3W_E_BAD_ARGUMENT; }
      nbits[i] = 1;
    } else if (folded && folded[i]) { ctx->last_error = "commit key: folded[] holds 0, 1 or 0x80 | bit"; return B3W_E_BAD_ARGUMENT; }
    first_v[i] = (uint32_t)nv;
    nv += nbits[i];
  }
  // window width: the caller's, else B3W_COMMIT_WINDOW, else 16 when its table takes at most a quarter of the free HBM
  DeviceGuard guard(ctx->device);
  hipError_t e = guard.err;
  uint32_t window = window_bits;
  if (!window && getenv("B3W_COMMIT_WINDOW")) {
    window = (uint32_t)atoi(getenv("B3W_COMMIT_WINDOW"));
    if (window != B3W_COMMIT_WINDOW_SMALL && window != B3W_COMMIT_WINDOW_LARGE) window = 0;
  }
  if (!window) {
    size_t free_b = 0, total_b = 0;
    const uint64_t large = (nv / B3W_COMMIT_WINDOW_LARGE + 1) * B3W_COMMIT_ENTRIES(B3W_COMMIT_WINDOW_LARGE) * 64;
    window = e == hipSuccess && hipMemGetInfo(&free_b, &total_b) == hipSuccess && large <= free_b / 4 ? B3W_COMMIT_WINDOW_LARGE
                                                                                                      : B3W_COMMIT_WINDOW_SMALL;
  }
  // windows of `window` virtual slots; the pad bits of the last window are never set
  const uint32_t V0 = (uint32_t)nv;
  nv = (nv + window - 1) / window * window;
  b3w_commit_key *key = new b3w_commit_key;
  key->ctx = ctx;
  key->curve = make_curve(curve == B3W_CURVE_BN254_G1 ? Q_BN254 : P_VESTA_BASE);
  key->first_slot = first_slot;
  key->window = window;
  key->nwin = (uint32_t)(nv / window);
  key->nslots = nslots;
  uint32_t *d_gens = nullptr, *d_first = nullptr, *d_nbits = nullptr, *d_points = nullptr;
  std::vector<uint32_t> desc(nslots);
  auto one_bit_of_word = [&](uint32_t i) { return folded && (folded[i] & 0x80) != 0; };
  for (uint32_t i = 0; i < nslots; i++)                  // code 4: folded, skipped; 8 + b: bit b of a 32-bit word
    desc[i] = first_v[i] | (one_bit_of_word(i) ? 8u + (folded[i] & 31u) : nbits[i] == 0 ? 4u : nbits[i] == 1 ? 0u : nbits[i] == 32 ? 1u : nbits[i] == 64 ? 2u : 3u) << 24;
  // O2 nova circuits: the 67 IsZero inverses of a step are 1/k of small signed k the step's inputs determine — one tabulated point
  // each instead of sixteen windows (b3w_commit_invtab_kernel).  B3W_COMMIT_INVTAB=0 turns it off.  Records mode takes k from the
  // record; bodies mode reads the four inputs from the body, COMPARES the body's inverse slot with +-1/k from the context's scalar
  // table and takes the point only if they are equal (code 5: gadget number in the low bits; anything else goes through the
  // windows like any 256-bit slot).
  static const bool want_invtab = !(getenv("B3W_COMMIT_INVTAB") && !strcmp(getenv("B3W_COMMIT_INVTAB"), "0"));
  // ... and only for the curve whose group order IS the circuit's prime: the table holds (+-1/k) G with 1/k taken in the circuit's
  // field, which is the scalar p - 1/k G's windows would add up to only when scalars live in that field (a caller of
  // the C API who pairs nova_vesta with BN254 G1 gets the windows, like with B3W_COMMIT_INVTAB=0, and consistent points either way)
  const bool order_is_prime = (ctx->desc.prime == P_BN254 && curve == B3W_CURVE_BN254_G1) || (ctx->desc.prime == P_VESTA && curve == B3W_CURVE_PALLAS);
  const bool with_invtab = want_invtab && order_is_prime && ctx->desc.kind == B3W_KIND_NOVA_O2 && ctx->d_aux;
  std::vector<uint32_t> invmeta(2 * B3W_NOVA_ISZERO + 4, 0xFFFFFFFFu);
  if (with_invtab) {
    for (uint32_t i = 0; i < nslots; i++) {
      const uint32_t src = table[first_slot + i] & 0xFFFu;           // (O2: the wide atoms are the inverses, wide index = gadget)
      if (nbits[i] == 256 && src >= B3W_LDS_WIDE && (src - B3W_LDS_WIDE) % 8 == 0 && (src - B3W_LDS_WIDE) / 8 < B3W_NOVA_ISZERO)
        invmeta[(src - B3W_LDS_WIDE) / 8] = i;
    }
    const uint32_t want_src[4] = {B3W_LDS_NV + NV_N_BLOCKS, B3W_LDS_NV + NV_BLOCK_COUNT, B3W_LDS_NV + NV_TOTAL_DEPTH, B3W_LDS_NV + NV_DEPTH};
    bool inputs_found = true;
    for (int q = 0; q < 4; q++) {
      for (uint32_t sidx = 0; sidx < ctx->desc.nwit && invmeta[B3W_NOVA_ISZERO + q] == 0xFFFFFFFFu; sidx++)
        if ((table[sidx] & 0xFFFu) == want_src[q] && ((table[sidx] >> 17) & 3u) == B3W_MODE_W32 && ((table[sidx] >> 12) & 31u) == 0) invmeta[B3W_NOVA_ISZERO + q] = sidx;
      inputs_found = inputs_found && invmeta[B3W_NOVA_ISZERO + q] != 0xFFFFFFFFu;
    }
    for (uint32_t j = 0; j < B3W_NOVA_ISZERO; j++)
      if (invmeta[j] != 0xFFFFFFFFu) {
        invmeta[B3W_NOVA_ISZERO + 4 + j] = first_v[invmeta[j]];
        if (inputs_found) desc[invmeta[j]] = j | 5u << 24;
      }
  }
  // records mode: slot s holds (image[src] >> sh) & mask (b3w_kernels.hip emit_group), so a run of bit slots reading
  // consecutive bits of one image word is one contiguous piece of the bit string
  std::vector<uint32_t> runs;
  for (uint32_t i = 0; i < nslots; i++) {
    const uint32_t ent = table[first_slot + i], src = ent & 0xFFFu, sh = (ent >> 12) & 31u, v0 = first_v[i];
    if (nbits[i] == 0) continue;
    if (one_bit_of_word(i)) {                               // one bit of the word's image word
      runs.push_back(v0 | 0u << 24); runs.push_back(src | (sh + (folded[i] & 31u)) << 16);
      continue;
    }
    if (nbits[i] == 1) {
      if (!runs.empty()) {
        const uint32_t a = runs[runs.size() - 2], b = runs[runs.size() - 1];
        const uint32_t plen = (a >> 24) + 1, pv = a & 0xFFFFFFu, psrc = b & 0xFFFFu, psh = b >> 16;
        if ((a >> 31) == 0 && psrc == src && psh + plen == sh && pv + plen == v0 && plen < 32 && i > 0 && nbits[i - 1] == 1) {
          runs[runs.size() - 2] = pv | plen << 24;               // one bit longer
          continue;
        }
      }
      runs.push_back(v0 | 0u << 24); runs.push_back(src | sh << 16);
    } else {
      const uint32_t words = nbits[i] / 32;                     // 1, 2 or 8 image words; the shift applies to words 0 and 4
      for (uint32_t k = 0; k < words; k++) {
        runs.push_back((v0 + 32 * k) | 31u << 24); runs.push_back((src + k) | ((k == 0 || k == 4) ? sh : 0u) << 16);
      }
    }
  }
  key->nruns = (uint32_t)(runs.size() / 2);
  if (e == hipSuccess) e = hipMalloc((void **)&key->d_runs, runs.size() * 4);
  if (e == hipSuccess) e = hipMemcpy(key->d_runs, runs.data(), runs.size() * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc((void **)&key->d_slotdesc, (size_t)nslots * 4);
  if (e == hipSuccess) e = hipMemcpy(key->d_slotdesc, desc.data(), (size_t)nslots * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc((void **)&key->d_table, (size_t)key->nwin * B3W_COMMIT_ENTRIES(window) * 64);
  if (e == hipSuccess) e = hipMalloc((void **)&d_points, (size_t)nv * 64);
  if (e == hipSuccess) e = hipMalloc((void **)&d_gens, (size_t)nslots * 64);
  if (e == hipSuccess) e = hipMalloc((void **)&d_first, (size_t)nslots * 4);
  if (e == hipSuccess) e = hipMalloc((void **)&d_nbits, (size_t)nslots * 4);
  if (e == hipSuccess) e = hipMemcpy(d_gens, host_generators, (size_t)nslots * 64, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_first, first_v.data(), (size_t)nslots * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_nbits, nbits.data(), (size_t)nslots * 4, hipMemcpyHostToDevice);
  // pad points: copies of the first point (never selected, but the table kernel adds them)
  int rc = e == hipSuccess ? b3w_launch_commit_setup(d_gens, d_first, d_nbits, nslots, d_points, &key->curve, nullptr) : 0;
  for (uint32_t v = V0; v < nv && e == hipSuccess && rc == 0; v++)
    e = hipMemcpyAsync(d_points + (size_t)v * 16, d_points, 64, hipMemcpyDeviceToDevice, nullptr);
  if (e == hipSuccess && rc == 0) rc = b3w_launch_commit_windows(d_points, key->nwin, key->window, key->d_table, &key->curve, nullptr);
  if (e == hipSuccess && rc == 0 && with_invtab) {       // (the tables themselves: see invmeta above)
    key->inv_nk = B3W_INV_TABLE_N - 1;
    e = hipMalloc((void **)&key->d_invmeta, invmeta.size() * 4);
    if (e == hipSuccess) e = hipMemcpy(key->d_invmeta, invmeta.data(), invmeta.size() * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void **)&key->d_invtab, (size_t)B3W_NOVA_ISZERO * 2 * key->inv_nk * 64);
    if (e == hipSuccess)
      rc = b3w_launch_commit_invtab(d_gens, key->d_invmeta, static_cast<const uint32_t *>(ctx->d_aux) + 16, B3W_NOVA_ISZERO, key->inv_nk, key->d_invtab, &key->curve,
                                    nullptr);
  }
  if (e == hipSuccess && rc == 0) e = hipDeviceSynchronize();
  if (d_gens) (void)hipFree(d_gens);
  if (d_first) (void)hipFree(d_first);
  if (d_nbits) (void)hipFree(d_nbits);
  if (d_points) (void)hipFree(d_points);
  if (e != hipSuccess || rc != 0) {
    b3w_commit_key_destroy(key);
    return hip_fail(ctx, e != hipSuccess ? e : (hipError_t)rc, "commitment key set-up");
  }
  *out = key;
  return B3W_OK;
}

void b3w_commit_key_destroy(b3w_commit_key *key) {
  if (!key) return;
  DeviceGuard guard(key->ctx->device);
  if (key->d_slotdesc) (void)hipFree(key->d_slotdesc);
  if (key->d_invmeta) (void)hipFree(key->d_invmeta);
  if (key->d_runs) (void)hipFree(key->d_runs);
  if (key->d_images) (void)hipFree(key->d_images);
  if (key->d_table) (void)hipFree(key->d_table);
  if (key->d_invtab) (void)hipFree(key->d_invtab);
  if (key->d_sums) (void)hipFree(key->d_sums);
  if (key->d_counts) (void)hipFree(key->d_counts);
  delete key;
}

int32_t b3w_commit_key_count(b3w_commit_key *key, int32_t on) {
  if (!key) return B3W_E_BAD_ARGUMENT;
  b3w_ctx *ctx = key->ctx;
  ON_DEVICE(ctx);
  HIP_TRY(ctx, hipDeviceSynchronize());
  if (on && !key->d_counts) HIP_TRY(ctx, hipMalloc((void **)&key->d_counts, 8));
  if (on) { HIP_TRY(ctx, hipMemset(key->d_counts, 0, 8)); key->host_witnesses = 0; }
  key->counting = on != 0;
  return B3W_OK;
}

int32_t b3w_commit_key_counts(const b3w_commit_key *key, uint64_t out[2]) {
  if (!key || !out) return B3W_E_BAD_ARGUMENT;
  b3w_ctx *ctx = key->ctx;
  out[0] = 0; out[1] = key->host_witnesses;
  if (!key->d_counts) return B3W_OK;
  ON_DEVICE(ctx);
  HIP_TRY(ctx, hipDeviceSynchronize());
  unsigned long long v = 0;
  HIP_TRY(ctx, hipMemcpy(&v, key->d_counts, 8, hipMemcpyDeviceToHost));
  out[0] = v;
  return B3W_OK;
}

int32_t b3w_batch_commit_device(b3w_ctx *ctx, const b3w_commit_key *key, const uint8_t *d_bodies, uint32_t n, uint64_t pitch,
                                uint8_t *d_points, int32_t *d_status, void *stream) {
  if (!ctx || !key || key->ctx != ctx || !d_bodies || !d_points) return B3W_E_BAD_ARGUMENT;
  const uint64_t body = 32ull * ctx->desc.nwit;
  if (pitch == 0) pitch = body;
  if (pitch < body || (pitch & 15) || (reinterpret_cast<uintptr_t>(d_bodies) & 15) || (reinterpret_cast<uintptr_t>(d_points) & 15)) {
    ctx->last_error = "pitch must be >= witness_size*32 and a multiple of 16, bodies and points 16-byte aligned";
    return B3W_E_BAD_ARGUMENT;
  }
  b3w_commit_key *k = const_cast<b3w_commit_key *>(key);                     // scratch only
  ON_DEVICE(ctx);
  if (k->sums_cap < n) {
    HIP_TRY(ctx, hipStreamSynchronize((hipStream_t)stream));
    if (k->d_sums) (void)hipFree(k->d_sums);
    k->d_sums = nullptr; k->sums_cap = 0;
    HIP_TRY(ctx, hipMalloc((void **)&k->d_sums, (size_t)n * B3W_COMMIT_SUM_WORDS * 4));
    k->sums_cap = n;
  }
  const int rc = b3w_launch_commit(d_bodies, n, pitch, key->first_slot, key->nslots, key->d_slotdesc, nullptr, 0, nullptr, 0, key->d_table,
                                   key->nwin, key->window, k->d_sums, d_points, d_status, key->d_invtab, key->inv_nk, key->d_invmeta,
                                   static_cast<const uint32_t *>(ctx->d_aux), key->counting ? key->d_counts : nullptr, &key->curve, (hipStream_t)stream, 0);
  if (rc == 0 && key->counting) k->host_witnesses += n;
  return rc ? hip_fail(ctx, (hipError_t)rc, "commit launch") : B3W_OK;
}

}  // extern "C"

// co_resident: the commit kernel built to share the device with the witness kernel of the same steps (the chained pass, GATED / FREE)
static int32_t commit_records_impl(b3w_ctx *ctx, const b3w_commit_key *key, const uint32_t *d_records, uint32_t n, uint8_t *d_points,
                                   uint32_t *d_public, int32_t *d_status, void *stream, bool co_resident) {
  if (!ctx || !key || key->ctx != ctx || !d_records || !d_points || !d_status) return B3W_E_BAD_ARGUMENT;
  if (n == 0) return B3W_OK;
  if ((reinterpret_cast<uintptr_t>(d_points) & 15) || (reinterpret_cast<uintptr_t>(d_records) & 3)) {
    ctx->last_error = "records 4-byte and points 16-byte aligned";
    return B3W_E_BAD_ARGUMENT;
  }
  ON_DEVICE(ctx);
  b3w_commit_key *k = const_cast<b3w_commit_key *>(key);                     // scratch only
  constexpr uint32_t CHUNK = 32768;                                          // witnesses per TRACE + commit pair (images: 3.7-11 KB each)
  const uint32_t want = std::min(n, CHUNK);
  if (k->images_cap < want || k->sums_cap < want) {
    HIP_TRY(ctx, hipStreamSynchronize((hipStream_t)stream));
    if (k->sums_cap < want) {
      if (k->d_sums) (void)hipFree(k->d_sums);
      k->d_sums = nullptr; k->sums_cap = 0;
      HIP_TRY(ctx, hipMalloc((void **)&k->d_sums, (size_t)want * B3W_COMMIT_SUM_WORDS * 4));
      k->sums_cap = want;
    }
    if (k->images_cap < want) {
      if (k->d_images) (void)hipFree(k->d_images);
      k->d_images = nullptr; k->images_cap = 0;
      HIP_TRY(ctx, hipMalloc((void **)&k->d_images, (size_t)want * ctx->desc.lds_words * 4));
      k->images_cap = want;
    }
  }
  const uint32_t cap = k->images_cap;
  const uint32_t rw = ctx->desc.nin, pw = ctx->desc.npub;
  for (uint32_t c0 = 0; c0 < n; c0 += cap) {                                 // the TRACE images of one chunk at a time
    const uint32_t cn = std::min(cap, n - c0);
    int lrc = b3w_launch_trace(ctx->desc.kind, d_records + (uint64_t)c0 * rw, cn, k->d_images, cap, ctx->d_table, ctx->desc.nwit,
                               d_public ? d_public + (uint64_t)c0 * pw : nullptr, d_status + c0, ctx->d_aux, (hipStream_t)stream);
    if (lrc == 0)
      lrc = b3w_launch_commit(nullptr, cn, 0, key->first_slot, key->nslots, key->d_slotdesc, k->d_images, cap, key->d_runs, key->nruns,
                              key->d_table, key->nwin, key->window, k->d_sums, d_points + (uint64_t)c0 * 64, nullptr, key->d_invtab, key->inv_nk,
                              nullptr, nullptr, key->counting ? key->d_counts : nullptr, &key->curve, (hipStream_t)stream, co_resident ? 1 : 0);
    if (lrc == 0 && key->counting) k->host_witnesses += cn;
    if (lrc) return hip_fail(ctx, (hipError_t)lrc, "commit-from-records launch");
  }
  return B3W_OK;
}

extern "C" {

int32_t b3w_commit_records_device(b3w_ctx *ctx, const b3w_commit_key *key, const uint32_t *d_records, uint32_t n, uint8_t *d_points,
                                  uint32_t *d_public, int32_t *d_status, void *stream) {
  static const bool co = getenv("B3W_COMMIT_CO") && !strcmp(getenv("B3W_COMMIT_CO"), "1");     // (measurements: tools/ubench/overlap_commit_probe.py)
  return commit_records_impl(ctx, key, d_records, n, d_points, d_public, d_status, stream, co);
}

int32_t b3w_commit_records(b3w_ctx *ctx, const b3w_commit_key *key, const uint32_t *host_records, uint32_t n, uint8_t *host_points,
                           uint32_t *host_public, int32_t *host_status) {
  if (!ctx || !key || key->ctx != ctx || !host_records || !host_points) return B3W_E_BAD_ARGUMENT;
  if (n == 0) return B3W_OK;
  ON_DEVICE(ctx);
  const size_t rb = (size_t)n * ctx->desc.nin * 4, pb = (size_t)n * ctx->desc.npub * 4;
  uint8_t *d = nullptr;                                   // records | points | public outputs | status
  const size_t o_pts = (rb + 255) & ~(size_t)255, o_pub = o_pts + (size_t)n * 64, o_st = o_pub + ((pb + 255) & ~(size_t)255);
  HIP_TRY(ctx, hipMalloc((void **)&d, o_st + (size_t)n * 4));
  hipError_t e = hipMemset(d + o_pub, 0, o_st - o_pub);   // a rejected record's public outputs are not written: zeros
  if (e == hipSuccess) e = hipMemcpy(d, host_records, rb, hipMemcpyHostToDevice);
  int32_t rc = B3W_OK;
  if (e == hipSuccess) rc = b3w_commit_records_device(ctx, key, reinterpret_cast<uint32_t *>(d), n, d + o_pts, reinterpret_cast<uint32_t *>(d + o_pub),
                                                      reinterpret_cast<int32_t *>(d + o_st), nullptr);
  if (e == hipSuccess && rc == B3W_OK) e = hipMemcpy(host_points, d + o_pts, (size_t)n * 64, hipMemcpyDeviceToHost);
  if (e == hipSuccess && rc == B3W_OK && host_public) e = hipMemcpy(host_public, d + o_pub, pb, hipMemcpyDeviceToHost);
  if (e == hipSuccess && rc == B3W_OK && host_status) e = hipMemcpy(host_status, d + o_st, (size_t)n * 4, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (rc) return rc;
  return e == hipSuccess ? B3W_OK : hip_fail(ctx, e, "hipMemcpy(commit from records)");
}

void b3w_commit_consumer(void *user, const uint8_t *d_bodies, uint64_t pitch, uint64_t first_step, uint32_t count, void *stream) {
  b3w_commit_sink *sink = static_cast<b3w_commit_sink *>(user);
  if (!sink || !sink->ctx || !sink->key || !sink->d_points) return;
  const int32_t rc = b3w_batch_commit_device(sink->ctx, sink->key, d_bodies, count, pitch, sink->d_points + first_step * 64,
                                             sink->d_status ? sink->d_status + first_step : nullptr, stream);
  if (rc && !sink->error) sink->error = rc;
}

int32_t b3w_batch_commit(b3w_batch *b, const b3w_commit_key *key, uint8_t *host_points, int32_t *host_status) {
  if (!b || !key || !host_points) return B3W_E_BAD_ARGUMENT;
  b3w_ctx *ctx = b->ctx;
  if (!b->n) return B3W_OK;
  ON_DEVICE(ctx);
  uint8_t *d_pts = nullptr;
  int32_t *d_st = nullptr;
  HIP_TRY(ctx, hipMalloc((void **)&d_pts, (size_t)b->n * 64));
  HIP_TRY(ctx, hipMalloc((void **)&d_st, (size_t)b->n * 4));
  int32_t rc = b3w_batch_commit_device(ctx, key, b->d_bodies, b->n, b->pitch, d_pts, d_st, nullptr);
  hipError_t e = rc == B3W_OK ? hipMemcpy(host_points, d_pts, (size_t)b->n * 64, hipMemcpyDeviceToHost) : hipSuccess;
  if (rc == B3W_OK && e == hipSuccess && host_status) e = hipMemcpy(host_status, d_st, (size_t)b->n * 4, hipMemcpyDeviceToHost);
  (void)hipFree(d_pts); (void)hipFree(d_st);
  if (rc) return rc;
  return e == hipSuccess ? B3W_OK : hip_fail(ctx, e, "hipMemcpy(commitments)");
}

// ---------------------------------------------------------------- multi-GPU exchange (RCCL, loaded at run time)
}  // extern "C"

#include <dlfcn.h>

#include "b3w_hostcomm.h"

struct b3w_comm {
  b3w_ctx *ctx = nullptr;
  int32_t rank = 0, nranks = 1;
  enum Kind { RCCL, HOST, EXTERNAL } kind = RCCL;
  void *comm = nullptr;          // RCCL: ncclComm_t
  B3wHostComm *host = nullptr;   // HOST: the shared-memory segment, and two pinned staging buffers that grow with the messages
  uint8_t *h_send = nullptr, *h_recv = nullptr;
  uint64_t h_cap = 0;
  b3w_allgather_fn fn = nullptr; // EXTERNAL: the caller's collective
  void *user = nullptr;
};

namespace {
struct RcclId { char b[128]; };  // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128, passed by value)
struct Rccl {                    // the five entry points used, with rccl.h's signatures
  void *so = nullptr;
  int (*GetUniqueId)(void *id) = nullptr;
  int (*CommInitRank)(void **comm, int nranks, RcclId id, int rank) = nullptr;
  int (*AllGather)(const void *send, void *recv, size_t count, int dtype, void *comm, hipStream_t stream) = nullptr;
  int (*CommDestroy)(void *comm) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  std::string err;
} rccl;

void load_rccl_once() {
  void *so = nullptr;
  for (const char *name : {"librccl.so", "librccl.so.1"}) if (!so) so = dlopen(name, RTLD_NOW | RTLD_NOLOAD);   // one already in the process
  for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) if (!so) so = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
  if (!so) { rccl.err = std::string("cannot load librccl: ") + dlerror(); return; }
  rccl.GetUniqueId = (decltype(rccl.GetUniqueId))dlsym(so, "ncclGetUniqueId");
  rccl.CommInitRank = (decltype(rccl.CommInitRank))dlsym(so, "ncclCommInitRank");
  rccl.AllGather = (decltype(rccl.AllGather))dlsym(so, "ncclAllGather");
  rccl.CommDestroy = (decltype(rccl.CommDestroy))dlsym(so, "ncclCommDestroy");
  rccl.GetErrorString = (decltype(rccl.GetErrorString))dlsym(so, "ncclGetErrorString");
  if (!rccl.GetUniqueId || !rccl.CommInitRank || !rccl.AllGather || !rccl.CommDestroy || !rccl.GetErrorString) { rccl.err = "librccl lacks an ncclAllGather entry point"; return; }
  rccl.so = so;
}
std::once_flag rccl_once;
bool load_rccl() {                 // thread-safe: distinct contexts may create communicators from different threads
  std::call_once(rccl_once, load_rccl_once);
  return rccl.so != nullptr;
}
}  // namespace

extern "C" {

int32_t b3w_comm_unique_id(uint8_t id[B3W_COMM_ID_BYTES]) {
  if (!id) return B3W_E_BAD_ARGUMENT;
  if (!load_rccl()) return B3W_E_RCCL;
  return rccl.GetUniqueId(id) == 0 ? B3W_OK : B3W_E_RCCL;
}

int32_t b3w_comm_create(b3w_ctx *ctx, const uint8_t id[B3W_COMM_ID_BYTES], int32_t rank, int32_t nranks, b3w_comm **out) {
  if (!ctx || !id || !out || nranks < 1 || rank < 0 || rank >= nranks) return B3W_E_BAD_ARGUMENT;
  *out = nullptr;
  if (!load_rccl()) { ctx->last_error = rccl.err; return B3W_E_RCCL; }
  ON_DEVICE(ctx);
  RcclId uid;
  memcpy(uid.b, id, 128);
  void *comm = nullptr;
  const int rc = rccl.CommInitRank(&comm, nranks, uid, rank);
  if (rc != 0) { ctx->last_error = std::string("ncclCommInitRank: ") + rccl.GetErrorString(rc); return B3W_E_RCCL; }
  b3w_comm *c = new b3w_comm;
  c->ctx = ctx; c->comm = comm; c->rank = rank; c->nranks = nranks;
  *out = c;
  return B3W_OK;
}

int32_t b3w_comm_create_host(b3w_ctx *ctx, const char *name, int32_t rank, int32_t nranks, b3w_comm **out) {
  if (!ctx || !name || !out || nranks < 1 || rank < 0 || rank >= nranks) return B3W_E_BAD_ARGUMENT;
  *out = nullptr;
  const char *t = getenv("B3W_HOSTCOMM_TIMEOUT_S");
  char err[256] = "";
  B3wHostComm *hc = nullptr;
  // 4 MiB per rank at a time: config 4's exchanges (256 KiB of h_out per rank at two ranks) go through in one piece
  if (b3w_hostcomm_open(name, rank, nranks, 4u << 20, t ? atof(t) : 120.0, &hc, err, sizeof err) != 0) {
    ctx->last_error = err;
    return B3W_E_RCCL;
  }
  b3w_comm *c = new b3w_comm;
  c->ctx = ctx; c->rank = rank; c->nranks = nranks; c->kind = b3w_comm::HOST; c->host = hc;
  *out = c;
  return B3W_OK;
}

int32_t b3w_comm_create_external(b3w_ctx *ctx, int32_t rank, int32_t nranks, b3w_allgather_fn allgather, void *user, b3w_comm **out) {
  if (!ctx || !allgather || !out || nranks < 1 || rank < 0 || rank >= nranks) return B3W_E_BAD_ARGUMENT;
  b3w_comm *c = new b3w_comm;
  c->ctx = ctx; c->rank = rank; c->nranks = nranks; c->kind = b3w_comm::EXTERNAL; c->fn = allgather; c->user = user;
  *out = c;
  return B3W_OK;
}

int32_t b3w_comm_rank(const b3w_comm *c) { return c ? c->rank : -1; }
int32_t b3w_comm_size(const b3w_comm *c) { return c ? c->nranks : 0; }

void b3w_comm_destroy(b3w_comm *c) {
  if (!c) return;
  DeviceGuard guard(c->ctx->device);
  if (c->comm && rccl.CommDestroy) (void)rccl.CommDestroy(c->comm);
  if (c->h_send) (void)hipHostFree(c->h_send);
  if (c->h_recv) (void)hipHostFree(c->h_recv);
  b3w_hostcomm_close(c->host);
  delete c;
}

namespace {
// HOST transport: device -> pinned host -> shared-memory all-gather -> device, ordered on `stream` by waiting for it (twice)
int32_t host_allgather(b3w_comm *c, const void *d_send, void *d_recv, uint64_t bytes, hipStream_t st) {
  b3w_ctx *ctx = c->ctx;
  if (bytes > c->h_cap) {
    if (c->h_send) (void)hipHostFree(c->h_send);
    if (c->h_recv) (void)hipHostFree(c->h_recv);
    c->h_send = c->h_recv = nullptr; c->h_cap = 0;
    HIP_TRY(ctx, hipHostMalloc((void **)&c->h_send, bytes, hipHostMallocDefault));
    HIP_TRY(ctx, hipHostMalloc((void **)&c->h_recv, bytes * (uint64_t)c->nranks, hipHostMallocDefault));
    c->h_cap = bytes;
  }
  HIP_TRY(ctx, hipMemcpyAsync(c->h_send, d_send, bytes, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  char err[256] = "";
  if (b3w_hostcomm_allgather(c->host, c->h_send, c->h_recv, bytes, err, sizeof err) != 0) { ctx->last_error = err; return B3W_E_RCCL; }
  HIP_TRY(ctx, hipMemcpyAsync(d_recv, c->h_recv, bytes * (uint64_t)c->nranks, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));                     // the staging buffer is free for the next call, whatever its stream
  return B3W_OK;
}
}  // namespace

int32_t b3w_comm_allgather(b3w_comm *c, const void *d_send, void *d_recv, uint64_t bytes_per_rank, void *stream) {
  if (!c || !d_send || !d_recv || !bytes_per_rank) return B3W_E_BAD_ARGUMENT;
  ON_DEVICE(c->ctx);
  if (c->kind == b3w_comm::HOST) return host_allgather(c, d_send, d_recv, bytes_per_rank, (hipStream_t)stream);
  if (c->kind == b3w_comm::EXTERNAL) {
    const int32_t rc = c->fn(c->user, d_send, d_recv, bytes_per_rank, stream);
    if (rc != 0) {
      c->ctx->last_error = "the caller's all-gather (b3w_comm_create_external) returned " + std::to_string(rc);
      return B3W_E_RCCL;
    }
    return B3W_OK;
  }
  const int rc = rccl.AllGather(d_send, d_recv, (size_t)bytes_per_rank, /* ncclInt8 */ 0, c->comm, (hipStream_t)stream);
  if (rc != 0) { c->ctx->last_error = std::string("ncclAllGather: ") + rccl.GetErrorString(rc); return B3W_E_RCCL; }
  return B3W_OK;
}

int32_t b3w_batch_allgather_public(b3w_batch *b, b3w_comm *c, uint32_t *host_all) {
  if (!b || !c || !host_all || !b->n) return B3W_E_BAD_ARGUMENT;
  b3w_ctx *ctx = b->ctx;
  ON_DEVICE(ctx);
  const uint64_t per = (uint64_t)b->n * ctx->desc.npub * 4;
  void *d_all = nullptr;
  HIP_TRY(ctx, hipMalloc(&d_all, per * c->nranks));
  int32_t rc = b3w_comm_allgather(c, b->d_pub, d_all, per, nullptr);
  hipError_t e = rc == B3W_OK ? hipMemcpy(host_all, d_all, per * c->nranks, hipMemcpyDeviceToHost) : hipSuccess;
  (void)hipFree(d_all);
  if (rc) return rc;
  return e == hipSuccess ? B3W_OK : hip_fail(ctx, e, "hipMemcpy(gathered public outputs)");
}

// ---------------------------------------------------------------- chained mode: native driver
}  // extern "C"

struct b3w_chain {
  b3w_ctx *ctx = nullptr;
  uint64_t len = 0, n_chunks = 0, first_chunk = 0, n_leaf = 0, n_par = 0, nbatch = 0;
  uint32_t nl = 0, P = 0, last_blocks = 16, batch_steps = 0, ring = 0;
  bool has_last = false, complete = false, with_parents = false, cvs_in_levels = false;
  int32_t placement = B3W_PLACEMENT_PLAIN;
  uint8_t *d_pre = nullptr;
  uint32_t *d_recs = nullptr, *d_cvs = nullptr, *d_pub = nullptr, *d_levels = nullptr, *d_root = nullptr;
  int32_t *d_status = nullptr;
  std::vector<void *> bodies;
  hipStream_t copy = nullptr, side = nullptr;        // H2D slices; tree + parent planning beside the leaf witness kernels
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_cvs = nullptr, ev_par = nullptr;     // chunk CVs complete (main stream); parent records ready (side stream)
  const b3w_commit_key *co_key = nullptr;            // commitments from the step records, one point per step into co_points ...
  bool co_bodies = false;                            // ... instead of the bodies (false), or beside them (true: b3w_chain_commit_from_records)
  hipStream_t co_stream = nullptr;                   // beside them = on a stream of its own: the commit kernels are bound by the vector ALUs, the
  hipEvent_t ev_co_in = nullptr, ev_co_out = nullptr;//   witness kernels by HBM writes — they run side by side
  int32_t co_overlap = B3W_COMMIT_OVERLAP_AUTO;      // b3w_chain_commit_overlap: where those commitments run
  int32_t *d_co_scratch = nullptr;                   // the side-stream commit kernel's status words: the witness kernel of the same records is
                                                     // the one that reports (d_status), this is never read
  uint8_t *co_points = nullptr, *co_own = nullptr;   // (co_own: the chain's own buffer when the caller passed none)
  const b3w_r1cs *r1cs = nullptr;                    // constraint check of every batch while it sits in the ring
  uint32_t *d_viol = nullptr;                        // ... violated constraints per step
  // sharded passes: exchange buffers, allocated on the first exchange for that communicator's rank count and kept
  // (no allocation, no host synchronisation inside a pass that has run once)
  struct Exchange {
    int32_t nranks = 0;
    uint64_t mx_chunks = 0, mx_leaf = 0, mx_par = 0;  // largest shard: chunks, leaf steps, parent steps
    uint32_t *d_cv_pad = nullptr, *d_cv_gath = nullptr, *d_cv_all = nullptr;
    uint32_t *d_h_send = nullptr, *d_h_recv = nullptr;
    uint64_t *d_tab = nullptr;                        // per rank {leaf dst row, leaf rows, parent dst row, parent rows}
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};   // around the chunk-CV exchange and around the h_out exchange (b3w_chain_exchange_ms)
    bool timed[2] = {false, false};
  } x;
};

namespace {
// Preimage per H2D slice = leaf steps per plan + witness (+ consumer) round: 1 MiB (16 384 steps) for small preimages, so that the
// copy of slice i + 1 hides under slice i; an eighth of the local preimage, up to 8 MiB, for large ones — fewer, larger launches and
// longer stretches in which the commit stream runs beside the witness kernels (64 MiB, profiles/r04/chain_slice_chunks.log: none 9.32
// -> 9.41, check 4.19 -> 4.22, commit 4.22 -> 4.44 M steps/s).  B3W_CHAIN_SLICE_CHUNKS overrides.
constexpr uint32_t CHAIN_SLICE_CHUNKS = 1024, CHAIN_SLICE_CHUNKS_MAX = 8192;
constexpr uint64_t RING_SPARE_CAP = 26ull << 30;    // ring buffers a context keeps between chains: two 16 384-step nova buffers

// roctx ranges around the stages of the chained pass (H2D slice, leaf planning, witness batches, consumer, tree + parent
// planning): `rocprofv3 --marker-trace --kernel-trace --memory-copy-trace` then shows which kernels and copies belong to
// which stage and how they overlap (profiles/r02/chain_*).  The marker library is looked up at run time, and only under a profiler (or
// B3W_ROCTX=1); otherwise a range costs one branch.
struct Roctx {
  int (*push)(const char *) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    // only under a profiler (rocprofv3 exports ROCP_TOOL_LIBRARIES) or when asked for: B3W_ROCTX=1
    const char *want = getenv("B3W_ROCTX");
    if (want ? strcmp(want, "1") != 0 : getenv("ROCP_TOOL_LIBRARIES") == nullptr) return;
    void *so = nullptr;
    for (const char *name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"})
      if (!so) so = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (!so) return;
    push = (int (*)(const char *))dlsym(so, "roctxRangePushA");
    pop = (int (*)())dlsym(so, "roctxRangePop");
    if (!push || !pop) { push = nullptr; pop = nullptr; }
  }
};
Roctx &roctx() { static Roctx r; return r; }
struct Range {
  bool on;
  explicit Range(const char *name) : on(roctx().push != nullptr) { if (on) roctx().push(name); }
  ~Range() { if (on) roctx().pop(); }
};

// where the commitments of a batch run (b3w_chain_commit_overlap); needs co_stream for anything but SERIAL
int32_t chain_commit_mode(const b3w_chain *c, bool has_consumer) {
  if (!(c->co_key && c->co_bodies && c->co_stream)) return B3W_COMMIT_OVERLAP_SERIAL;
  if (c->co_overlap != B3W_COMMIT_OVERLAP_AUTO) return c->co_overlap;
  // something reads the batch on `stream` right after the witness kernel (the constraint check: every VGPR and 138 KB of LDS per CU):
  // the commitments overlap the witness kernel only.  Nothing does: they run free beside the witness kernels of this and later batches.
  return (c->r1cs || has_consumer) ? B3W_COMMIT_OVERLAP_GATED : B3W_COMMIT_OVERLAP_FREE;
}

int32_t chain_run_steps(b3w_chain *c, uint64_t first_row, uint64_t count, b3w_batch_consumer consumer, void *user, void *stream) {
  const uint64_t body = 32ull * c->ctx->desc.nwit;
  const int32_t mode = chain_commit_mode(c, consumer != nullptr);
  for (uint64_t done = 0; done < count;) {
    const uint32_t k = (uint32_t)std::min<uint64_t>(c->batch_steps, count - done);
    uint8_t *slot = static_cast<uint8_t *>(c->bodies[c->nbatch % c->ring]);
    const uint64_t r0 = first_row + done;
    if (mode != B3W_COMMIT_OVERLAP_SERIAL) {
      // beside the bodies: on the commit stream, behind everything `stream` holds so far (the records of this batch are planned; GATED:
      // the check of the previous batch is over)
      Range r("b3w:commit from records (side stream)");
      hipError_t e = hipEventRecord(c->ev_co_in, (hipStream_t)stream);
      if (e == hipSuccess) e = hipStreamWaitEvent(c->co_stream, c->ev_co_in, 0);
      if (e != hipSuccess) return hip_fail(c->ctx, e, "commit stream");
      const int32_t rc = b3w_commit_records_device(c->ctx, c->co_key, c->d_recs + r0 * 32, k, c->co_points + r0 * 64, nullptr, c->d_co_scratch + r0, c->co_stream);
      if (rc) return rc;
      if (mode == B3W_COMMIT_OVERLAP_GATED && (e = hipEventRecord(c->ev_co_out, c->co_stream)) != hipSuccess) return hip_fail(c->ctx, e, "commit stream");
    } else if (c->co_key) {
      Range r("b3w:commit from records");
      const int32_t rc = b3w_commit_records_device(c->ctx, c->co_key, c->d_recs + r0 * 32, k, c->co_points + r0 * 64, c->d_pub + r0 * 15,
                                                   c->d_status + r0, stream);
      if (rc) return rc;
      if (!c->co_bodies) {
        c->nbatch++;
        done += k;
        continue;
      }
    }
    int32_t rc;
    { Range r("b3w:witness batch"); rc = b3w_batch_run_device(c->ctx, c->d_recs + r0 * 32, k, slot, body, c->d_pub + r0 * 15, c->d_status + r0, stream); }
    if (rc) return rc;
    if (mode == B3W_COMMIT_OVERLAP_GATED) {                    // what reads the batch starts when BOTH are done: it gets the machine to itself
      const hipError_t e = hipStreamWaitEvent((hipStream_t)stream, c->ev_co_out, 0);
      if (e != hipSuccess) return hip_fail(c->ctx, e, "commit stream");
    }
    if (c->r1cs) {
      Range r("b3w:constraint check");
      rc = b3w_r1cs_check_device(c->ctx, c->r1cs, slot, k, body, c->d_viol + r0, nullptr, stream);
      if (rc) return rc;
    }
    if (consumer) { Range r("b3w:consumer"); consumer(user, slot, body, r0, k, stream); }
    c->nbatch++;
    done += k;
  }
  if (mode == B3W_COMMIT_OVERLAP_FREE) {                       // `stream` has drained = the commitments are there too
    hipError_t e = hipEventRecord(c->ev_co_out, c->co_stream);
    if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)stream, c->ev_co_out, 0);
    if (e != hipSuccess) return hip_fail(c->ctx, e, "commit stream");
  }
  return B3W_OK;
}
}  // namespace

extern "C" {

namespace {
void chain_drop_commit_stream(b3w_chain *c) {
  if (c->co_stream) { (void)hipStreamSynchronize(c->co_stream); (void)hipStreamDestroy(c->co_stream); c->co_stream = nullptr; }
  if (c->ev_co_in) { (void)hipEventDestroy(c->ev_co_in); c->ev_co_in = nullptr; }
  if (c->ev_co_out) { (void)hipEventDestroy(c->ev_co_out); c->ev_co_out = nullptr; }
}
}  // namespace

int32_t b3w_chain_commit_from_records(b3w_chain *c, const b3w_commit_key *key, uint8_t *d_points) {
  const int32_t rc = b3w_chain_commit_only(c, key, d_points);
  if (rc != B3W_OK) return rc;
  c->co_bodies = key != nullptr;
  // The commit stream.  Where the commitments run is b3w_chain_commit_overlap's choice (include/b3wit.h); SERIAL needs no stream.
  // B3W_COMMIT_CU_PCT=<p>: the stream may use only p % of the CUs (hipExtStreamCreateWithCUMask; measured slower at 75 and 88),
  // B3W_COMMIT_PRIORITY=<n>: its priority (hipStreamCreateWithPriority; measurement switches).
  const bool want_stream = key && c->co_overlap != B3W_COMMIT_OVERLAP_SERIAL;
  if (!want_stream && c->co_stream) {                        // (switched off for a chain that had it)
    ON_DEVICE(c->ctx);
    chain_drop_commit_stream(c);
  }
  if (want_stream && !c->co_stream) {
    b3w_ctx *ctx = c->ctx;
    ON_DEVICE(ctx);
    static const int pct = getenv("B3W_COMMIT_CU_PCT") ? atoi(getenv("B3W_COMMIT_CU_PCT")) : 0;
    static const char *prio = getenv("B3W_COMMIT_PRIORITY");
    hipError_t e = hipSuccess;
    if (pct > 0 && pct < 100) {
      int cus = 0;
      e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
      std::vector<uint32_t> mask((size_t)(cus + 31) / 32, 0u);
      // every (100 / (100 - pct))-th CU stays out of the mask: spread over the XCDs (CU ids are dealt to them round-robin)
      for (int cu = 0; cu < cus; cu++) if ((int64_t)cu * (100 - pct) / 100 == (int64_t)(cu + 1) * (100 - pct) / 100) mask[cu / 32] |= 1u << (cu % 32);
      if (e == hipSuccess) e = hipExtStreamCreateWithCUMask(&c->co_stream, (uint32_t)mask.size(), mask.data());
    } else if (prio) e = hipStreamCreateWithPriority(&c->co_stream, hipStreamNonBlocking, atoi(prio));
    else e = hipStreamCreateWithFlags(&c->co_stream, hipStreamNonBlocking);
    if (e == hipSuccess && !c->ev_co_in) e = hipEventCreateWithFlags(&c->ev_co_in, hipEventDisableTiming);
    if (e == hipSuccess && !c->ev_co_out) e = hipEventCreateWithFlags(&c->ev_co_out, hipEventDisableTiming);
    if (e == hipSuccess && !c->d_co_scratch) e = hipMalloc((void **)&c->d_co_scratch, (size_t)(c->n_leaf + c->n_par + 1) * 4);
    if (e != hipSuccess) {                                   // no half-built stream: the chain is back to "no commitments" and says why
      chain_drop_commit_stream(c);
      c->co_key = nullptr; c->co_points = nullptr; c->co_bodies = false;
      return hip_fail(ctx, e, "commit stream");
    }
  }
  return B3W_OK;
}

int32_t b3w_chain_commit_overlap(b3w_chain *c, int32_t mode) {
  if (!c || mode < B3W_COMMIT_OVERLAP_AUTO || mode > B3W_COMMIT_OVERLAP_GATED) return B3W_E_BAD_ARGUMENT;
  c->co_overlap = mode;
  if (c->co_key && c->co_bodies) return b3w_chain_commit_from_records(c, c->co_key, c->co_points);   // (stream made or dropped to match)
  return B3W_OK;
}

uint32_t *b3w_chain_violations_device(b3w_chain *c) { return c ? c->d_viol : nullptr; }

int32_t b3w_chain_commit_only(b3w_chain *c, const b3w_commit_key *key, uint8_t *d_points) {
  if (!c || (key && key->ctx != c->ctx)) return B3W_E_BAD_ARGUMENT;
  c->co_bodies = false;
  if (key && !d_points) {                              // the chain's own buffer: fetch it with b3w_chain_commitments
    if (!c->co_own) {
      ON_DEVICE(c->ctx);
      HIP_TRY(c->ctx, hipMalloc((void **)&c->co_own, (size_t)(c->n_leaf + c->n_par + 1) * 64));
    }
    d_points = c->co_own;
  }
  c->co_key = key;
  c->co_points = key ? d_points : nullptr;
  return B3W_OK;
}

int32_t b3w_chain_check_constraints(b3w_chain *c, const b3w_r1cs *r1cs) {
  if (!c || (r1cs && r1cs->ctx != c->ctx)) return B3W_E_BAD_ARGUMENT;
  if (r1cs && !c->d_viol) {
    ON_DEVICE(c->ctx);
    HIP_TRY(c->ctx, hipMalloc((void **)&c->d_viol, (size_t)(c->n_leaf + c->n_par + 1) * 4));
  }
  c->r1cs = r1cs;
  return B3W_OK;
}

int32_t b3w_chain_violations(b3w_chain *c, uint32_t *host_violations, void *stream) {
  if (!c || !host_violations || !c->r1cs || !c->d_viol) return B3W_E_BAD_ARGUMENT;
  ON_DEVICE(c->ctx);
  HIP_TRY(c->ctx, hipStreamSynchronize((hipStream_t)stream));
  HIP_TRY(c->ctx, hipMemcpy(host_violations, c->d_viol, (size_t)(c->n_leaf + c->n_par) * 4, hipMemcpyDeviceToHost));
  return B3W_OK;
}

int32_t b3w_chain_commitments(b3w_chain *c, uint8_t *host_points, void *stream) {
  if (!c || !host_points || !c->co_points) return B3W_E_BAD_ARGUMENT;
  ON_DEVICE(c->ctx);
  HIP_TRY(c->ctx, hipStreamSynchronize((hipStream_t)stream));
  HIP_TRY(c->ctx, hipMemcpy(host_points, c->co_points, (size_t)(c->n_leaf + c->n_par) * 64, hipMemcpyDeviceToHost));
  return B3W_OK;
}

int32_t b3w_chain_create(b3w_ctx *ctx, uint64_t preimage_len, uint64_t first_chunk, uint32_t n_chunks_local, uint32_t batch_steps,
                         uint32_t ring, int32_t with_parents, b3w_chain **out) {
  if (!ctx || !out || !batch_steps || !ring) return B3W_E_BAD_ARGUMENT;
  *out = nullptr;
  if (ctx->desc.kind == B3W_KIND_COMP) { ctx->last_error = "chained mode runs the nova step circuits"; return B3W_E_BAD_ARGUMENT; }
  const uint64_t n = b3w_chain_num_chunks(preimage_len);
  if (first_chunk + n_chunks_local > n) { ctx->last_error = "chunk range exceeds the preimage"; return B3W_E_BAD_ARGUMENT; }
  b3w_chain *c = new b3w_chain;
  c->ctx = ctx; c->len = preimage_len; c->n_chunks = n; c->first_chunk = first_chunk; c->nl = n_chunks_local;
  c->batch_steps = batch_steps; c->ring = ring;
  c->P = b3w_plan_path_len(0, n);
  c->complete = (n & (n - 1)) == 0;
  const uint64_t last_bytes = preimage_len > (n - 1) * 1024 ? preimage_len - (n - 1) * 1024 : 0;
  c->last_blocks = last_bytes ? (uint32_t)((last_bytes + 63) / 64) : 1;
  c->has_last = first_chunk + n_chunks_local == n;
  c->n_leaf = (uint64_t)n_chunks_local * 16 - ((c->has_last && n_chunks_local) ? 16 - c->last_blocks : 0);
  c->with_parents = with_parents != 0;
  c->n_par = with_parents ? b3w_chain_num_parent_steps(preimage_len, first_chunk, n_chunks_local) : 0;
  const uint64_t rows = (uint64_t)n_chunks_local * 16 + c->n_par + 1;
  const uint64_t body = 32ull * ctx->desc.nwit;
  DeviceGuard guard(ctx->device);
  hipError_t e = guard.err;
  if (e == hipSuccess) e = hipMalloc((void **)&c->d_pre, std::max<uint64_t>(n_chunks_local, 1) * 1024);
  if (e == hipSuccess) e = hipMalloc((void **)&c->d_recs, rows * 32 * 4);
  c->cvs_in_levels = n_chunks_local == n;              // all chunks here: the leaf planner writes the chunk CVs straight into level 0 of the tree
  if (e == hipSuccess && !c->cvs_in_levels) e = hipMalloc((void **)&c->d_cvs, std::max<uint64_t>(n_chunks_local, 1) * 8 * 4);
  if (e == hipSuccess) e = hipMalloc((void **)&c->d_pub, rows * 15 * 4);
  if (e == hipSuccess) e = hipMalloc((void **)&c->d_status, rows * 4);
  if (e == hipSuccess) e = hipMalloc((void **)&c->d_levels, (2 * n + 64) * 8 * 4);
  if (e == hipSuccess && c->cvs_in_levels) c->d_cvs = c->d_levels;
  if (e == hipSuccess) e = hipMalloc((void **)&c->d_root, 8 * 4);
  if (e == hipSuccess) e = hipMemset(c->d_status, 0, rows * 4);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->copy, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_cvs, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_par, hipEventDisableTiming);
  for (int i = 0; i < 4 && e == hipSuccess; i++) e = hipEventCreateWithFlags(&c->ev[i], hipEventDisableTiming);
  if (e != hipSuccess) { b3w_chain_destroy(c); return e == hipErrorOutOfMemory ? B3W_E_NOT_ENOUGH_MEMORY : hip_fail(ctx, e, "chain buffers"); }
  c->placement = B3W_PLACEMENT_MIXED;
  if (!ctx->ring_spares.empty() && ctx->ring_spares[0].bytes != (uint64_t)batch_steps * body) (void)b3w_ctx_trim(ctx);   // spares of another geometry: evicted
  for (uint32_t i = 0; i < ring; i++) {
    void *p = nullptr;
    int32_t pl = B3W_PLACEMENT_PLAIN;
    const uint64_t want = (uint64_t)batch_steps * body;
    for (size_t k = 0; k < ctx->ring_spares.size() && !p; k++)         // a ring buffer of an earlier chain of this context
      if (ctx->ring_spares[k].bytes == want) {
        p = ctx->ring_spares[k].ptr; pl = ctx->ring_spares[k].placement;
        ctx->ring_spares.erase(ctx->ring_spares.begin() + k);
      }
    const int32_t rc = p ? B3W_OK : b3w_bodies_alloc(ctx, want, &p, &pl);
    if (rc) { b3w_chain_destroy(c); return rc; }
    c->bodies.push_back(p);
    if (pl == B3W_PLACEMENT_PLAIN) c->placement = B3W_PLACEMENT_PLAIN;                    // the weakest of the ring's buffers names the ring
    else if (pl == B3W_PLACEMENT_INTERLEAVED && c->placement == B3W_PLACEMENT_MIXED) c->placement = B3W_PLACEMENT_INTERLEAVED;
  }
  *out = c;
  return B3W_OK;
}

void b3w_chain_destroy(b3w_chain *c) {
  if (!c) return;
  DeviceGuard guard(c->ctx->device);
  (void)hipDeviceSynchronize();
  // ring buffers go back to the context (placed buffers use up address space for good: DESIGN.md "Placement") for the next
  // chain of the same ring geometry.  Spares of another size are released first (one size at a time) and the spares never
  // hold more than RING_SPARE_CAP bytes; b3w_ctx_trim releases them.
  const uint64_t ring_bytes = (uint64_t)c->batch_steps * 32ull * c->ctx->desc.nwit;
  {
    std::vector<b3w_ctx::Spare> &sp = c->ctx->ring_spares;
    if (!sp.empty() && sp[0].bytes != ring_bytes) (void)b3w_ctx_trim(c->ctx);
    uint64_t held = (uint64_t)sp.size() * ring_bytes;
    for (void *p : c->bodies) {
      if (held + ring_bytes <= RING_SPARE_CAP) { sp.push_back({p, ring_bytes, c->placement}); held += ring_bytes; }
      else (void)b3w_bodies_free(c->ctx, p);
    }
  }
  for (void *q : {(void *)c->x.d_cv_pad, (void *)c->x.d_cv_gath, (void *)c->x.d_cv_all, (void *)c->x.d_h_send, (void *)c->x.d_h_recv, (void *)c->x.d_tab})
    if (q) (void)hipFree(q);
  for (hipEvent_t e : c->x.ev) if (e) (void)hipEventDestroy(e);
  if (c->d_pre) (void)hipFree(c->d_pre);
  if (c->d_recs) (void)hipFree(c->d_recs);
  if (c->d_cvs && !c->cvs_in_levels) (void)hipFree(c->d_cvs);
  if (c->d_pub) (void)hipFree(c->d_pub);
  if (c->d_status) (void)hipFree(c->d_status);
  if (c->d_levels) (void)hipFree(c->d_levels);
  if (c->d_root) (void)hipFree(c->d_root);
  if (c->co_own) (void)hipFree(c->co_own);
  if (c->d_co_scratch) (void)hipFree(c->d_co_scratch);
  chain_drop_commit_stream(c);
  if (c->d_viol) (void)hipFree(c->d_viol);
  if (c->copy) (void)hipStreamDestroy(c->copy);
  if (c->side) (void)hipStreamDestroy(c->side);
  if (c->ev_cvs) (void)hipEventDestroy(c->ev_cvs);
  if (c->ev_par) (void)hipEventDestroy(c->ev_par);
  for (int i = 0; i < 4; i++) if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
  delete c;
}

int32_t b3w_chain_run_leaves(b3w_chain *c, const uint8_t *host_preimage, b3w_batch_consumer consumer, void *user, void *stream) {
  if (!c || !host_preimage) return B3W_E_BAD_ARGUMENT;
  b3w_ctx *ctx = c->ctx;
  ON_DEVICE(ctx);
  hipStream_t st = (hipStream_t)stream;
  // the copy stream must not run ahead of work still reading d_pre from an earlier pass on `stream`
  HIP_TRY(ctx, hipEventRecord(c->ev[3], st));
  HIP_TRY(ctx, hipStreamWaitEvent(c->copy, c->ev[3], 0));
  uint32_t slice = 0;
  static const uint32_t slice_env = getenv("B3W_CHAIN_SLICE_CHUNKS") ? (uint32_t)std::max(1, atoi(getenv("B3W_CHAIN_SLICE_CHUNKS"))) : 0u;
  const uint32_t SLICE = slice_env ? slice_env : std::min<uint32_t>(CHAIN_SLICE_CHUNKS_MAX, std::max<uint32_t>(CHAIN_SLICE_CHUNKS, (uint32_t)(c->nl / 8)));
  for (uint32_t s0 = 0; s0 < c->nl; s0 += SLICE, slice++) {
    const uint32_t sc = std::min<uint32_t>(SLICE, c->nl - s0);
    const uint64_t b0 = (c->first_chunk + s0) * 1024, b1 = std::min<uint64_t>(b0 + (uint64_t)sc * 1024, c->len);
    hipEvent_t ev = c->ev[slice % 3];
    {
      Range r("b3w:h2d preimage slice");
      if (b1 > b0) HIP_TRY(ctx, hipMemcpyAsync(c->d_pre + (uint64_t)s0 * 1024, host_preimage + b0, b1 - b0, hipMemcpyHostToDevice, c->copy));
      HIP_TRY(ctx, hipEventRecord(ev, c->copy));
      HIP_TRY(ctx, hipStreamWaitEvent(st, ev, 0));
    }
    int32_t rc;
    {
      Range r("b3w:plan leaf steps");
      rc = b3w_chain_plan_leaves_device(ctx, c->d_pre + (uint64_t)s0 * 1024, c->len, c->first_chunk + s0, sc,
                                        c->d_recs + (uint64_t)s0 * 16 * 32, c->d_cvs + (uint64_t)s0 * 8, stream);
    }
    if (rc) return rc;
    if (s0 + sc == c->nl) HIP_TRY(ctx, hipEventRecord(c->ev_cvs, st));     // every local chunk CV is on its way
    const uint64_t steps_here = (uint64_t)sc * 16 - ((c->has_last && s0 + sc == c->nl) ? 16 - c->last_blocks : 0);
    rc = chain_run_steps(c, (uint64_t)s0 * 16, steps_here, consumer, user, stream);
    if (rc) return rc;
  }
  return B3W_OK;
}

namespace {
// cvs_on_side: d_all_chunk_cvs was written by work already queued on the chain's side stream (the sharded pass's own exchange)
int32_t chain_run_parents(b3w_chain *c, const uint32_t *d_all_chunk_cvs, bool cvs_on_side, b3w_batch_consumer consumer, void *user, void *stream) {
  b3w_ctx *ctx = c->ctx;
  hipStream_t st = (hipStream_t)stream;
  // The tree and the parent-step records only need the chunk CVs, not the leaf witnesses: they run on a side stream
  // beside the leaf witness kernels still queued on `stream` (250 us of small dependent launches for a 1 MiB preimage).
  if (!d_all_chunk_cvs) {
    if (c->nl != c->n_chunks) { ctx->last_error = "a chunk sub-range needs the chunk CVs of all ranks"; return B3W_E_BAD_ARGUMENT; }
    d_all_chunk_cvs = c->d_cvs;
    HIP_TRY(ctx, hipStreamWaitEvent(c->side, c->ev_cvs, 0));
  } else if (!cvs_on_side) {                           // gathered by the caller on `stream`: order after that
    HIP_TRY(ctx, hipEventRecord(c->ev_cvs, st));
    HIP_TRY(ctx, hipStreamWaitEvent(c->side, c->ev_cvs, 0));
  }
  int32_t rc;
  {
    Range r("b3w:tree + plan parent steps");
    if (d_all_chunk_cvs != c->d_levels)                // (a single rank plans its chunk CVs into level 0, an even sharded pass gathers them there)
      HIP_TRY(ctx, hipMemcpyAsync(c->d_levels, d_all_chunk_cvs, c->n_chunks * 32, hipMemcpyDeviceToDevice, c->side));
    // few local chunks (a rank's share of a small preimage): the tree kernel's workgroup plans their parent steps itself; many: the
    // path kernel's workgroups, spread over the chip, behind it
    const bool fused_plan = c->n_par && c->n_chunks > 1 && c->nl <= 256;
    rc = chain_tree(ctx, c->d_levels, c->n_chunks, c->d_root, c->first_chunk, fused_plan ? c->nl : 0, c->last_blocks, c->d_recs + c->n_leaf * 32, c->side);
    if (rc == B3W_OK && c->n_par && !fused_plan)
      rc = b3w_chain_plan_parents_device(ctx, c->d_levels, c->n_chunks, c->len, c->first_chunk, c->nl, c->d_recs + c->n_leaf * 32, c->side);
  }
  HIP_TRY(ctx, hipEventRecord(c->ev_par, c->side));
  HIP_TRY(ctx, hipStreamWaitEvent(st, c->ev_par, 0));   // also when something failed: `stream` must not run ahead of the side stream
  if (rc) return rc;
  if (!c->n_par) return B3W_OK;
  return chain_run_steps(c, c->n_leaf, c->n_par, consumer, user, stream);
}
}  // namespace

int32_t b3w_chain_run_parents(b3w_chain *c, const uint32_t *d_all_chunk_cvs, b3w_batch_consumer consumer, void *user, void *stream) {
  if (!c) return B3W_E_BAD_ARGUMENT;
  ON_DEVICE(c->ctx);
  return chain_run_parents(c, d_all_chunk_cvs, false, consumer, user, stream);
}

void b3w_chain_shard(uint64_t n_chunks, int32_t rank, int32_t nranks, uint64_t *first_chunk, uint32_t *n_chunks_local) {
  if (nranks < 1) nranks = 1;
  const uint64_t q = n_chunks / (uint64_t)nranks, r = n_chunks % (uint64_t)nranks, k = (uint64_t)(rank < 0 ? 0 : rank);
  if (first_chunk) *first_chunk = k * q + (k < r ? k : r);
  if (n_chunks_local) *n_chunks_local = (uint32_t)(q + (k < r ? 1 : 0));
}

namespace {
// exchange buffers of a sharded pass, sized for `nranks` (allocated once per chain; a communicator of another size re-allocates)
int32_t chain_exchange(b3w_chain *c, int32_t nranks) {
  b3w_chain::Exchange &x = c->x;
  if (x.nranks == nranks) return B3W_OK;
  b3w_ctx *ctx = c->ctx;
  for (void *q : {(void *)x.d_cv_pad, (void *)x.d_cv_gath, (void *)x.d_cv_all, (void *)x.d_h_send, (void *)x.d_h_recv, (void *)x.d_tab})
    if (q) (void)hipFree(q);
  for (hipEvent_t ev : x.ev) if (ev) (void)hipEventDestroy(ev);
  x = b3w_chain::Exchange();
  std::vector<uint64_t> tab(4 * (size_t)nranks);
  uint64_t mxc = 0, mxl = 0, mxp = 0;
  const uint64_t last_short = 16 - c->last_blocks;          // steps the last chunk of the preimage lacks
  for (int32_t r = 0; r < nranks; r++) {
    uint64_t f = 0; uint32_t k = 0;
    b3w_chain_shard(c->n_chunks, r, nranks, &f, &k);
    const uint64_t leaf = (uint64_t)k * 16 - ((k && f + k == c->n_chunks) ? last_short : 0);
    const uint64_t p0 = c->with_parents ? b3w_plan_parent_row(f, c->n_chunks) : 0, p1 = c->with_parents ? b3w_plan_parent_row(f + k, c->n_chunks) : 0;
    tab[4 * r] = f * 16; tab[4 * r + 1] = leaf; tab[4 * r + 2] = p0; tab[4 * r + 3] = p1 - p0;
    mxc = std::max<uint64_t>(mxc, k); mxl = std::max(mxl, leaf); mxp = std::max(mxp, p1 - p0);
  }
  x.mx_chunks = std::max<uint64_t>(mxc, 1); x.mx_leaf = std::max<uint64_t>(mxl, 1); x.mx_par = mxp;
  const uint64_t hwords = (x.mx_leaf + x.mx_par) * 8;
  hipError_t e = hipMalloc((void **)&x.d_cv_pad, x.mx_chunks * 32);
  if (e == hipSuccess) e = hipMalloc((void **)&x.d_cv_gath, x.mx_chunks * 32 * nranks);
  if (e == hipSuccess) e = hipMalloc((void **)&x.d_cv_all, c->n_chunks * 32);
  if (e == hipSuccess) e = hipMalloc((void **)&x.d_h_send, hwords * 4);
  if (e == hipSuccess) e = hipMalloc((void **)&x.d_h_recv, hwords * 4 * nranks);
  if (e == hipSuccess) e = hipMalloc((void **)&x.d_tab, tab.size() * 8);
  if (e == hipSuccess) e = hipMemset(x.d_cv_pad, 0, x.mx_chunks * 32);           // the padding goes over the wire: zeros, once
  if (e == hipSuccess) e = hipMemset(x.d_h_send, 0, hwords * 4);
  if (e == hipSuccess) e = hipMemcpy(x.d_tab, tab.data(), tab.size() * 8, hipMemcpyHostToDevice);
  for (int i = 0; i < 4 && e == hipSuccess; i++) e = hipEventCreate(&x.ev[i]);
  if (e != hipSuccess) return e == hipErrorOutOfMemory ? B3W_E_NOT_ENOUGH_MEMORY : hip_fail(ctx, e, "exchange buffers of the sharded pass");
  x.nranks = nranks;
  return B3W_OK;
}

int32_t chain_check_shard(b3w_chain *c, const b3w_comm *comm) {
  uint64_t first = 0; uint32_t count = 0;
  b3w_chain_shard(c->n_chunks, comm->rank, comm->nranks, &first, &count);
  if (first != c->first_chunk || count != c->nl) { c->ctx->last_error = "the chain was not created with this rank's b3w_chain_shard range"; return B3W_E_BAD_ARGUMENT; }
  return B3W_OK;
}
}  // namespace

int32_t b3w_chain_run_parents_sharded(b3w_chain *c, b3w_comm *comm, b3w_batch_consumer consumer, void *user, void *stream) {
  if (!c || !comm) return B3W_E_BAD_ARGUMENT;
  b3w_ctx *ctx = c->ctx;
  int32_t rc = chain_check_shard(c, comm);
  if (rc) return rc;
  ON_DEVICE(ctx);
  if ((rc = chain_exchange(c, comm->nranks)) != B3W_OK) return rc;
  // The chunk CVs exist as soon as the last leaf PLAN has run (ev_cvs, b3w_chain_run_leaves) — long before the leaf witness kernels
  // queued behind it on `stream` have finished.  Their exchange therefore runs on the chain's side stream, beside those kernels, and
  // the tree and the parent plan follow it there: `stream` only joins for the parent witnesses.  (On `stream` itself the exchange and
  // 250 us of small dependent launches stood behind the last leaf witness: a rank's share of a 1 MiB pass took 1.1 ms at 8 ranks
  // against 0.47 ms for an ordinary pass over a preimage of the shard's size, profiles/r04/chain_scaling_model_1mib.json.)
  hipStream_t side = c->side;
  b3w_chain::Exchange &x = c->x;
  // equal shards (config 4: 1 024 chunks over 8 ranks): no padding, so the collective takes the chunk CVs where they lie and leaves
  // them in global chunk order — no staging copy, no compaction (each a hipMemcpyAsync of its own: 17 of them cost 0.2 ms at 8 ranks)
  const bool even = c->n_chunks % (uint64_t)comm->nranks == 0;
  const uint32_t *d_all = even ? c->d_levels : x.d_cv_all;        // (even: level 0 of the tree IS the receive buffer)
  hipError_t e = hipStreamWaitEvent(side, c->ev_cvs, 0);
  if (e == hipSuccess) e = hipEventRecord(x.ev[0], side);
  if (e == hipSuccess && c->nl && !even) e = hipMemcpyAsync(x.d_cv_pad, c->d_cvs, (uint64_t)c->nl * 32, hipMemcpyDeviceToDevice, side);
  rc = e == hipSuccess ? b3w_comm_allgather(comm, even ? c->d_cvs : x.d_cv_pad, even ? c->d_levels : x.d_cv_gath, x.mx_chunks * 32, side)
                       : hip_fail(ctx, e, "chunk CV staging");
  for (int32_t r = 0; r < comm->nranks && rc == B3W_OK && !even; r++) {        // drop the padding: global chunk order
    uint64_t f = 0; uint32_t k = 0;
    b3w_chain_shard(c->n_chunks, r, comm->nranks, &f, &k);
    if (k && (e = hipMemcpyAsync(x.d_cv_all + f * 8, x.d_cv_gath + (uint64_t)r * x.mx_chunks * 8, (uint64_t)k * 32, hipMemcpyDeviceToDevice, side)) != hipSuccess)
      rc = hip_fail(ctx, e, "chunk CV compaction");
  }
  if (rc == B3W_OK && (e = hipEventRecord(x.ev[1], side)) != hipSuccess) rc = hip_fail(ctx, e, "hipEventRecord");
  if (rc) {                                                  // `stream` must not run ahead of what the side stream still holds
    if (hipEventRecord(c->ev_par, side) == hipSuccess) (void)hipStreamWaitEvent((hipStream_t)stream, c->ev_par, 0);
    return rc;
  }
  x.timed[0] = true;
  return chain_run_parents(c, d_all, true, consumer, user, stream);
}

int32_t b3w_chain_allgather_hout(b3w_chain *c, b3w_comm *comm, uint32_t *d_leaf_hout, uint32_t *d_parent_hout, void *stream) {
  if (!c || !comm || (!d_leaf_hout && !d_parent_hout)) return B3W_E_BAD_ARGUMENT;
  b3w_ctx *ctx = c->ctx;
  int32_t rc = chain_check_shard(c, comm);
  if (rc) return rc;
  ON_DEVICE(ctx);
  if ((rc = chain_exchange(c, comm->nranks)) != B3W_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  b3w_chain::Exchange &x = c->x;
  // wire format per rank: [leaf h_out, mx_leaf rows | parent h_out, mx_par rows], 8 words a row
  HIP_TRY(ctx, hipEventRecord(x.ev[2], st));
  int e = b3w_launch_pack_hout(c->d_pub, 0, c->n_leaf, x.d_h_send, st);
  if (e == 0) e = b3w_launch_pack_hout(c->d_pub, c->n_leaf, c->n_par, x.d_h_send + x.mx_leaf * 8, st);
  if (e) return hip_fail(ctx, (hipError_t)e, "h_out packing");
  const uint64_t block_words = (x.mx_leaf + x.mx_par) * 8;
  if ((rc = b3w_comm_allgather(comm, x.d_h_send, x.d_h_recv, block_words * 4, stream)) != B3W_OK) return rc;
  e = b3w_launch_unpack_hout(x.d_h_recv, block_words, x.mx_leaf * 8, x.d_tab, (uint32_t)comm->nranks, x.mx_leaf + x.mx_par, d_leaf_hout,
                             x.mx_par ? d_parent_hout : nullptr, st);
  if (e) return hip_fail(ctx, (hipError_t)e, "h_out unpacking");
  HIP_TRY(ctx, hipEventRecord(x.ev[3], st));
  x.timed[1] = true;
  return B3W_OK;
}

int32_t b3w_chain_exchange_ms(b3w_chain *c, float out_ms[2]) {
  if (!c || !out_ms) return B3W_E_BAD_ARGUMENT;
  b3w_ctx *ctx = c->ctx;
  ON_DEVICE(ctx);
  for (int k = 0; k < 2; k++) {
    out_ms[k] = 0.0f;
    if (!c->x.timed[k]) continue;
    HIP_TRY(ctx, hipEventSynchronize(c->x.ev[2 * k + 1]));
    HIP_TRY(ctx, hipEventElapsedTime(&out_ms[k], c->x.ev[2 * k], c->x.ev[2 * k + 1]));
  }
  return B3W_OK;
}

int32_t b3w_chain_allgather_hout_host(b3w_chain *c, b3w_comm *comm, uint32_t *host_leaf_hout, uint32_t *host_parent_hout, void *stream) {
  if (!c || !comm || (!host_leaf_hout && !host_parent_hout)) return B3W_E_BAD_ARGUMENT;
  b3w_ctx *ctx = c->ctx;
  ON_DEVICE(ctx);
  const uint64_t n_leaf = b3w_chain_num_leaf_steps(c->len), n_par = c->with_parents ? b3w_plan_parent_row(c->n_chunks, c->n_chunks) : 0;
  uint32_t *d = nullptr;
  HIP_TRY(ctx, hipMalloc((void **)&d, (size_t)(n_leaf + n_par + 1) * 32));
  int32_t rc = b3w_chain_allgather_hout(c, comm, d, n_par ? d + n_leaf * 8 : nullptr, stream);
  hipError_t e = rc == B3W_OK ? hipStreamSynchronize((hipStream_t)stream) : hipSuccess;
  if (rc == B3W_OK && e == hipSuccess && host_leaf_hout) e = hipMemcpy(host_leaf_hout, d, (size_t)n_leaf * 32, hipMemcpyDeviceToHost);
  if (rc == B3W_OK && e == hipSuccess && host_parent_hout && n_par) e = hipMemcpy(host_parent_hout, d + n_leaf * 8, (size_t)n_par * 32, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (rc) return rc;
  return e == hipSuccess ? B3W_OK : hip_fail(ctx, e, "h_out exchange (host)");
}

int32_t b3w_chain_info(const b3w_chain *c, uint64_t *n_leaf_steps, uint64_t *n_parent_steps, uint64_t *n_chunks, uint32_t *path_len,
                       int32_t *placement) {
  if (!c) return B3W_E_BAD_ARGUMENT;
  if (n_leaf_steps) *n_leaf_steps = c->n_leaf;
  if (n_parent_steps) *n_parent_steps = c->n_par;
  if (n_chunks) *n_chunks = c->n_chunks;
  if (path_len) *path_len = c->P;
  if (placement) *placement = c->placement;
  return B3W_OK;
}
int32_t b3w_chain_outputs(b3w_chain *c, uint32_t *host_public, int32_t *host_status, uint32_t *host_root, void *stream) {
  if (!c) return B3W_E_BAD_ARGUMENT;
  b3w_ctx *ctx = c->ctx;
  ON_DEVICE(ctx);
  HIP_TRY(ctx, hipStreamSynchronize((hipStream_t)stream));
  const uint64_t rows = c->n_leaf + c->n_par;
  if (host_public) HIP_TRY(ctx, hipMemcpy(host_public, c->d_pub, rows * 15 * 4, hipMemcpyDeviceToHost));
  if (host_status) HIP_TRY(ctx, hipMemcpy(host_status, c->d_status, rows * 4, hipMemcpyDeviceToHost));
  if (host_root) HIP_TRY(ctx, hipMemcpy(host_root, c->d_root, 32, hipMemcpyDeviceToHost));
  return B3W_OK;
}
uint32_t *b3w_chain_records(b3w_chain *c) { return c ? c->d_recs : nullptr; }
uint32_t *b3w_chain_public(b3w_chain *c) { return c ? c->d_pub : nullptr; }
int32_t *b3w_chain_status(b3w_chain *c) { return c ? c->d_status : nullptr; }
uint32_t *b3w_chain_local_cvs(b3w_chain *c) { return c ? c->d_cvs : nullptr; }
uint32_t *b3w_chain_root(b3w_chain *c) { return c ? c->d_root : nullptr; }

}  // extern "C"
