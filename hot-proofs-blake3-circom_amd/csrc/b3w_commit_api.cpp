// b3w_commit_api.cpp — C-ABI part 5: commitment keys and the commit entry points (on-device consumer #2) over b3w_commit.hip.
#include "b3w_internal.h"


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
  if (!b3w_int_build_slot_table(ctx->desc, table, ctx->last_error)) return B3W_E_BAD_ARGUMENT;
  for (uint32_t i = 0; i < ctx->desc.nwit; i++) {
    const uint32_t mode = (table[i] >> 17) & 3u;
    out_bits[i] = mode == B3W_MODE_BIT ? 1 : mode == B3W_MODE_W32 ? 32 : mode == B3W_MODE_W64 ? 64 : 256;
  }
  return B3W_OK;
}

// One attempt with the window given, or chosen (window_bits 0: *auto_window tells which, *herr what the runtime said when it failed).
static int32_t commit_key_create_once(b3w_ctx *ctx, int32_t curve, uint32_t first_slot, const uint8_t *host_generators,
                                      const uint8_t *folded, uint32_t window_bits, b3w_commit_key **out, uint32_t *auto_window, hipError_t *herr);

// An automatic window is a guess from hipMemGetInfo's free figure at one moment: other ranks or processes sizing their keys on the
// same GPU, torch or the placement pool taking memory between the query and the allocation, or fragmentation can make the 25-51 GB
// table of 18 bits fail where 16 (a quarter of it) or 12 fit.  Then — and only for a window nobody asked for — the next narrower
// one is tried instead of returning B3W_E_HIP.
int32_t b3w_commit_key_create_folded(b3w_ctx *ctx, int32_t curve, uint32_t first_slot, const uint8_t *host_generators,
                                     const uint8_t *folded /* per committed slot, or null */, uint32_t window_bits, b3w_commit_key **out) {
  uint32_t chosen = 0;
  hipError_t herr = hipSuccess;
  int32_t rc = commit_key_create_once(ctx, curve, first_slot, host_generators, folded, window_bits, out, &chosen, &herr);
  while (rc == B3W_E_HIP && herr == hipErrorOutOfMemory && chosen > B3W_COMMIT_WINDOW_SMALL) {
    (void)hipGetLastError();                               // the failed allocation's sticky error
    const uint32_t narrower = chosen == B3W_COMMIT_WINDOW_XL ? B3W_COMMIT_WINDOW_LARGE : B3W_COMMIT_WINDOW_SMALL;
    uint32_t ignored = 0;
    herr = hipSuccess;
    rc = commit_key_create_once(ctx, curve, first_slot, host_generators, folded, narrower, out, &ignored, &herr);
    chosen = narrower;
  }
  return rc;
}

static int32_t commit_key_create_once(b3w_ctx *ctx, int32_t curve, uint32_t first_slot, const uint8_t *host_generators,
                                      const uint8_t *folded, uint32_t window_bits, b3w_commit_key **out, uint32_t *auto_window, hipError_t *herr) {
  if (!ctx || !out || !host_generators || (curve != B3W_CURVE_BN254_G1 && curve != B3W_CURVE_VESTA) || first_slot >= ctx->desc.nwit ||
      (window_bits != 0 && !B3W_COMMIT_WINDOW_OK(window_bits))) {
    if (ctx) ctx->last_error = "commit key: curve 0/1, first_slot < witness_size, window_bits 0 (auto), 12, 16 or 18";
    return B3W_E_BAD_ARGUMENT;
  }
  *out = nullptr;
  std::vector<uint32_t> table;
  if (!b3w_int_build_slot_table(ctx->desc, table, ctx->last_error)) return B3W_E_BAD_ARGUMENT;
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
  // window width: the caller's, else B3W_COMMIT_WINDOW, else the widest of 18, 16, 12 whose table takes at most a quarter of the free HBM
  DeviceGuard guard(ctx->device);
  hipError_t e = guard.err;
  uint32_t window = window_bits;
  if (!window && getenv("B3W_COMMIT_WINDOW")) {
    window = (uint32_t)atoi(getenv("B3W_COMMIT_WINDOW"));
    if (!B3W_COMMIT_WINDOW_OK(window)) window = 0;
  }
  if (!window) {
    size_t free_b = 0, total_b = 0;
    bool known = e == hipSuccess && hipMemGetInfo(&free_b, &total_b) == hipSuccess;
    // (tests of the fallback below: the figure the choice BELIEVES — as if the memory had been free at the query and gone at the allocation)
    if (known && getenv("B3W_COMMIT_ASSUME_FREE_GIB")) free_b = (size_t)atoll(getenv("B3W_COMMIT_ASSUME_FREE_GIB")) << 30;
    auto table_bytes = [&](uint32_t w) { return (nv / w + 1) * (uint64_t)B3W_COMMIT_ENTRIES(w) * 64; };
    window = known && table_bytes(B3W_COMMIT_WINDOW_XL) <= free_b / 4      ? B3W_COMMIT_WINDOW_XL
             : known && table_bytes(B3W_COMMIT_WINDOW_LARGE) <= free_b / 4 ? B3W_COMMIT_WINDOW_LARGE
                                                                           : B3W_COMMIT_WINDOW_SMALL;
    *auto_window = window;                               // nobody asked for this width: a failed allocation may fall back
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
    *herr = e != hipSuccess ? e : (hipError_t)rc;
    (void)hipGetLastError();                               // not left behind for the next launch's hipGetLastError() to find
    return hip_fail(ctx, e != hipSuccess ? e : (hipError_t)rc, "commitment key set-up");
  }
  *out = key;
  return B3W_OK;
}

void b3w_commit_key_destroy(b3w_commit_key *key) {
  if (!key) return;
  B3wCaptureRelaxed relaxed;                                 // (b3w_capture.h)
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
                                   static_cast<const uint32_t *>(ctx->d_aux), key->counting ? key->d_counts : nullptr, &key->curve, (hipStream_t)stream);
  if (rc == 0 && key->counting) k->host_witnesses += n;
  return rc ? hip_fail(ctx, (hipError_t)rc, "commit launch") : B3W_OK;
}

int32_t b3w_commit_records_device(b3w_ctx *ctx, const b3w_commit_key *key, const uint32_t *d_records, uint32_t n, uint8_t *d_points,
                                  uint32_t *d_public, int32_t *d_status, void *stream) {
  return b3w_int_commit_records(ctx, key, d_records, n, d_points, d_public, d_status, stream, nullptr, nullptr, nullptr);
}

}  // extern "C"

int32_t b3w_int_commit_normalize(b3w_ctx *ctx, const b3w_commit_key *key, const uint32_t *d_sums, uint64_t n, uint8_t *d_points, void *stream) {
  if (!ctx || !key || key->ctx != ctx || !d_sums || !d_points) return B3W_E_BAD_ARGUMENT;
  ON_DEVICE(ctx);
  const int rc = b3w_launch_commit_normalize(d_sums, n, d_points, &key->curve, (hipStream_t)stream);
  return rc ? hip_fail(ctx, (hipError_t)rc, "commit normalise launch") : B3W_OK;
}

int32_t b3w_int_commit_records(b3w_ctx *ctx, const b3w_commit_key *key, const uint32_t *d_records, uint32_t n, uint8_t *d_points, uint32_t *d_public,
                               int32_t *d_status, void *stream, uint32_t *d_sums_out, void *trace_stream, hipEvent_t trace_done) {
  if (!ctx || !key || key->ctx != ctx || !d_records || !d_points || !d_status) return B3W_E_BAD_ARGUMENT;
  const bool split = trace_done != nullptr && trace_stream != stream;         // (trace_stream may be the null stream: a stream like any other)
  if (split && n > 32768u) return B3W_E_BAD_ARGUMENT;                         // (one chunk: the next chunk's TRACE would have to wait for this one's commit kernel)
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
    if (split) HIP_TRY(ctx, hipStreamSynchronize((hipStream_t)trace_stream));
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
                               d_public ? d_public + (uint64_t)c0 * pw : nullptr, d_status + c0, ctx->d_aux, (hipStream_t)(split ? trace_stream : stream));
    if (lrc == 0 && split) {
      hipError_t e = hipEventRecord(trace_done, (hipStream_t)trace_stream);
      if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)stream, trace_done, 0);
      if (e != hipSuccess) return hip_fail(ctx, e, "TRACE event");
    }
    if (lrc == 0)
      lrc = b3w_launch_commit(nullptr, cn, 0, key->first_slot, key->nslots, key->d_slotdesc, k->d_images, cap, key->d_runs, key->nruns,
                              key->d_table, key->nwin, key->window, d_sums_out ? d_sums_out + (uint64_t)c0 * B3W_COMMIT_SUM_WORDS : k->d_sums,
                              d_sums_out ? nullptr : d_points + (uint64_t)c0 * 64, nullptr, key->d_invtab, key->inv_nk,
                              nullptr, nullptr, key->counting ? key->d_counts : nullptr, &key->curve, (hipStream_t)stream);
    if (lrc == 0 && key->counting) k->host_witnesses += cn;
    if (lrc) return hip_fail(ctx, (hipError_t)lrc, "commit-from-records launch");
  }
  return B3W_OK;
}

extern "C" {

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

}  // extern "C"

b3w_ctx *b3w_int_key_ctx(const b3w_commit_key *key) { return key ? key->ctx : nullptr; }
