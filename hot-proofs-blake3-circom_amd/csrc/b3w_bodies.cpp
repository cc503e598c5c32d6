// b3w_bodies.cpp — C-ABI part 2: body buffers (the placement allocator's front end and its real-kernel check) and the batch object
// that owns device buffers.
#include "b3w_internal.h"

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
    // B3W_PLACEMENT=single (harness): every piece from ONE class of memory — a worst-case plain buffer made on purpose, labelled plain
    const bool single = env && !strcmp(env, "single");
    const int rc = b3w_place_alloc(ctx->device, bytes, single ? 2 : 1, d_ptr, &mixed, nullptr);
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
  B3wCaptureRelaxed relaxed;                                 // (b3w_capture.h)
  int cur = 0;
  (void)hipGetDevice(&cur);
  DeviceGuard guard(ctx ? ctx->device : cur);
  if (b3w_place_free(d_ptr) == 0) return B3W_OK;
  hipError_t e = hipFree(d_ptr);
  return e == hipSuccess ? B3W_OK : hip_fail(ctx, e, "hipFree(bodies)");
}

void b3w_bodies_trim(void) { B3wCaptureRelaxed relaxed; b3w_place_trim(); }

int32_t b3w_ctx_trim(b3w_ctx *ctx) {
  if (!ctx) return B3W_E_BAD_ARGUMENT;
  B3wCaptureRelaxed relaxed;                                 // (b3w_capture.h)
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
  B3wCaptureRelaxed relaxed;                                 // (b3w_capture.h)
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
  // (a batch whose own buffer came out plain — no second class of memory found — gets the fill-ordered kernel from the default policy,
  // which asks the placement allocator what it knows about the buffer: b3w_int_default_variant)
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
void *b3w_batch_device_ptr(b3w_batch *b, uint64_t *pitch) {
  if (!b) return nullptr;
  if (pitch) *pitch = b->pitch;
  return b->d_bodies;
}

}  // extern "C"
