// b3w_r1cs_api.cpp — C-ABI part 4: the rank-1 constraint check (on-device consumer #1): the b3w_r1cs object over b3w_r1cs_host.cpp's
// programs and the kernels of b3w_r1cs_walk.hip / b3w_r1cs.hip.
#include "b3w_internal.h"


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
  B3wCaptureRelaxed relaxed;                                 // (b3w_capture.h)
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
  // the walk kernel where the system fits it, else the stream kernel, else the gather kernel.
  // B3W_R1CS_GATHER picks another formulation, for comparison: 1 the gather kernel, 4 the stream kernel (round 3's default)
  static const int other = getenv("B3W_R1CS_GATHER") ? atoi(getenv("B3W_R1CS_GATHER")) : 0;
  if (r->tiled && (other == 0 || other == 4)) {
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
    if (lrc == -6) lrc = b3w_launch_r1cs_stream(d_bodies, n, pitch, &r->sys, &r->field, scratch, d_violations, d_first, (hipStream_t)stream);
    if (lrc != -6) {
      if (done) (void)hipEventRecord(done, (hipStream_t)stream);
      return lrc ? hip_fail(ctx, (hipError_t)lrc, "r1cs check launch") : B3W_OK;
    }
    // (a tiled system neither kernel has room for: the gather kernel below takes any system)
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

}  // extern "C"

b3w_ctx *b3w_int_r1cs_ctx(const b3w_r1cs *r1cs) { return r1cs ? r1cs->ctx : nullptr; }
