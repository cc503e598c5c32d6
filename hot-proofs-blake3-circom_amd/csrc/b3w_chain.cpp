// b3w_chain.cpp — C-ABI part 7: chained ("nova fold") mode: the planner wrappers and the native driver of the whole pass
// (b3w_chain_*): preimage slices -> leaf plan -> witness batches through a ring of placed buffers -> tree + parent plan -> parent
// witnesses, with the constraint check, the commitments from records and the sharded exchanges.
#include "b3w_internal.h"
#include <dlfcn.h>

extern "C" {

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
  hipStream_t last_caller = nullptr;                 // the stream of the last run call (b3w_chain_destroy waits for it when it has to wait stream by stream)
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
  uint32_t *d_co_sums = nullptr;                     // the steps' projective sums where the points are made once per run call, not per batch (chain_run_steps)
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
  // Commitments ONLY (no bodies, no consumer call: nobody is promised a batch's points before the run call returns): the batches leave
  // their projective sums and ONE launch behind the last batch makes the points.  A batch's own normalisation is a chain of 380
  // dependent multiplications on one wave per CU — 130 us with nothing else running, 6 % of such a pass: 7.63 -> 8.04 M steps/s
  // (profiles/r05/commit_defer_normalize.log).  Beside bodies the per-batch launch hides behind the witness kernels and putting it
  // off gains nothing (FREE: 4.69 -> 4.62), and under SERIAL / GATED the batch's consumer may read its points: those keep theirs.
  static const bool defer_env = !(getenv("B3W_COMMIT_DEFER_NORMALIZE") && !strcmp(getenv("B3W_COMMIT_DEFER_NORMALIZE"), "0"));
  bool defer = defer_env && c->co_key && mode == B3W_COMMIT_OVERLAP_SERIAL && !c->co_bodies;
  if (defer && !c->d_co_sums &&                              // (144 bytes a step, the first commit-only pass of the chain; none to be had: the batches normalise their own)
      hipMalloc((void **)&c->d_co_sums, (size_t)(c->n_leaf + c->n_par + 1) * B3W_COMMIT_SUM_WORDS * 4) != hipSuccess) {
    (void)hipGetLastError();
    c->d_co_sums = nullptr;
    defer = false;
  }
  uint32_t *sums = defer ? c->d_co_sums : nullptr;
  for (uint64_t done = 0; done < count;) {
    const uint32_t k = (uint32_t)std::min<uint64_t>(c->batch_steps, count - done);
    uint8_t *slot = static_cast<uint8_t *>(c->bodies[c->nbatch % c->ring]);
    const uint64_t r0 = first_row + done;
    if (mode != B3W_COMMIT_OVERLAP_SERIAL) {
      // beside the bodies: on the commit stream, behind everything `stream` holds so far (the records of this batch are planned; GATED:
      // the check of the previous batch is over)
      Range r("b3w:commit from records (side stream)");
      // GATED: the TRACE images (115 MB of stores for 16 384 nova steps) are written on `stream`, IN FRONT of the witness kernel — 40 us
      // there, 640 us beside it, and the commit stream is the longer of the two (timeline_ranks1_8mib_check_commit.txt) —; the commit
      // kernel waits for them on its own stream (the event stands for ev_co_in: behind everything `stream` held, the last batch's
      // check included, which is also why the key's one image buffer is free again).  FREE keeps both on the commit stream: its
      // witness kernels run ahead of the commitments, and the image buffer of the batch before would still be read.
      static const bool split_env = !(getenv("B3W_COMMIT_SPLIT_TRACE") && !strcmp(getenv("B3W_COMMIT_SPLIT_TRACE"), "0"));
      const bool split = split_env && mode == B3W_COMMIT_OVERLAP_GATED && k <= 32768u;
      hipError_t e = hipSuccess;
      if (!split) {
        e = hipEventRecord(c->ev_co_in, (hipStream_t)stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(c->co_stream, c->ev_co_in, 0);
        if (e != hipSuccess) return hip_fail(c->ctx, e, "commit stream");
      }
      const int32_t rc = b3w_int_commit_records(c->ctx, c->co_key, c->d_recs + r0 * 32, k, c->co_points + r0 * 64, nullptr, c->d_co_scratch + r0, c->co_stream, nullptr,
                                                split ? stream : nullptr, split ? c->ev_co_in : nullptr);
      if (rc) return rc;
      if (mode == B3W_COMMIT_OVERLAP_GATED && (e = hipEventRecord(c->ev_co_out, c->co_stream)) != hipSuccess) return hip_fail(c->ctx, e, "commit stream");
    } else if (c->co_key) {
      Range r("b3w:commit from records");
      const int32_t rc = b3w_int_commit_records(c->ctx, c->co_key, c->d_recs + r0 * 32, k, c->co_points + r0 * 64, c->d_pub + r0 * 15,
                                                c->d_status + r0, stream, sums ? sums + r0 * B3W_COMMIT_SUM_WORDS : nullptr);
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
  if (defer && count) {                                        // the points of this call's steps, behind the last batch's sums on the stream that made them
    Range r("b3w:commit normalise");
    const int32_t rc = b3w_int_commit_normalize(c->ctx, c->co_key, sums + first_row * B3W_COMMIT_SUM_WORDS, count, c->co_points + first_row * 64, stream);
    if (rc) return rc;
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
  if (!c || (key && b3w_int_key_ctx(key) != c->ctx)) return B3W_E_BAD_ARGUMENT;
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
  if (!c || (r1cs && b3w_int_r1cs_ctx(r1cs) != c->ctx)) return B3W_E_BAD_ARGUMENT;
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
  B3wCaptureRelaxed relaxed;                                 // (b3w_capture.h)
  DeviceGuard guard(c->ctx->device);
  {                                                          // the ring buffers outlive the chain (spares): nothing may still write them
    const hipStream_t mine[4] = {c->last_caller, c->copy, c->side, c->co_stream};
    (void)b3w_device_wait(mine, 4);                          // (says so on stderr when it could not wait: b3w_capture.h)
  }
  // ring buffers go back to the context (placed buffers use up address space for good: DESIGN.md "Placement") for the next
  // chain of the same ring geometry.  Spares of another size are released first (one size at a time) and the spares never
  // hold more than B3W_RING_SPARE_CAP bytes; b3w_ctx_trim releases them.
  const uint64_t ring_bytes = (uint64_t)c->batch_steps * 32ull * c->ctx->desc.nwit;
  {
    std::vector<b3w_ctx::Spare> &sp = c->ctx->ring_spares;
    if (!sp.empty() && sp[0].bytes != ring_bytes) (void)b3w_ctx_trim(c->ctx);
    uint64_t held = (uint64_t)sp.size() * ring_bytes;
    for (void *p : c->bodies) {
      if (held + ring_bytes <= B3W_RING_SPARE_CAP) { sp.push_back({p, ring_bytes, c->placement}); held += ring_bytes; }
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
  if (c->d_co_sums) (void)hipFree(c->d_co_sums);
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
  if (c) c->last_caller = (hipStream_t)stream;
  if (!c || !host_preimage) return B3W_E_BAD_ARGUMENT;
  b3w_ctx *ctx = c->ctx;
  ON_DEVICE(ctx);
  hipStream_t st = (hipStream_t)stream;
  uint32_t slice = 0;
  static const uint32_t slice_env = getenv("B3W_CHAIN_SLICE_CHUNKS") ? (uint32_t)std::max(1, atoi(getenv("B3W_CHAIN_SLICE_CHUNKS"))) : 0u;
  const uint32_t SLICE = slice_env ? slice_env : std::min<uint32_t>(CHAIN_SLICE_CHUNKS_MAX, std::max<uint32_t>(CHAIN_SLICE_CHUNKS, (uint32_t)(c->nl / 8)));
  // ONE slice (a preimage of up to 1 MiB, a rank's share of config 4): nothing to overlap the copy with, so it goes on `stream` itself —
  // an event from the copy stream to `stream` is 20-30 us of latency in front of the first kernel (profiles/r05/timeline_*.txt)
  const bool one_slice = c->nl <= SLICE;
  if (!one_slice) {
    // the copy stream must not run ahead of work still reading d_pre from an earlier pass on `stream`
    HIP_TRY(ctx, hipEventRecord(c->ev[3], st));
    HIP_TRY(ctx, hipStreamWaitEvent(c->copy, c->ev[3], 0));
  }
  for (uint32_t s0 = 0; s0 < c->nl; s0 += SLICE, slice++) {
    const uint32_t sc = std::min<uint32_t>(SLICE, c->nl - s0);
    const uint64_t b0 = (c->first_chunk + s0) * 1024, b1 = std::min<uint64_t>(b0 + (uint64_t)sc * 1024, c->len);
    hipEvent_t ev = c->ev[slice % 3];
    {
      Range r("b3w:h2d preimage slice");
      if (b1 > b0) HIP_TRY(ctx, hipMemcpyAsync(c->d_pre + (uint64_t)s0 * 1024, host_preimage + b0, b1 - b0, hipMemcpyHostToDevice, one_slice ? st : c->copy));
      if (!one_slice) {
        HIP_TRY(ctx, hipEventRecord(ev, c->copy));
        HIP_TRY(ctx, hipStreamWaitEvent(st, ev, 0));
      }
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
  if (c) c->last_caller = (hipStream_t)stream;
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
  b3w_chain_shard(c->n_chunks, b3w_comm_rank(comm), b3w_comm_size(comm), &first, &count);
  if (first != c->first_chunk || count != c->nl) { c->ctx->last_error = "the chain was not created with this rank's b3w_chain_shard range"; return B3W_E_BAD_ARGUMENT; }
  return B3W_OK;
}
}  // namespace

int32_t b3w_chain_run_parents_sharded(b3w_chain *c, b3w_comm *comm, b3w_batch_consumer consumer, void *user, void *stream) {
  if (c) c->last_caller = (hipStream_t)stream;
  if (!c || !comm) return B3W_E_BAD_ARGUMENT;
  b3w_ctx *ctx = c->ctx;
  int32_t rc = chain_check_shard(c, comm);
  if (rc) return rc;
  ON_DEVICE(ctx);
  if ((rc = chain_exchange(c, b3w_comm_size(comm))) != B3W_OK) return rc;
  // The chunk CVs exist as soon as the last leaf PLAN has run (ev_cvs, b3w_chain_run_leaves) — long before the leaf witness kernels
  // queued behind it on `stream` have finished.  Their exchange therefore runs on the chain's side stream, beside those kernels, and
  // the tree and the parent plan follow it there: `stream` only joins for the parent witnesses.  (On `stream` itself the exchange and
  // 250 us of small dependent launches stood behind the last leaf witness: a rank's share of a 1 MiB pass took 1.1 ms at 8 ranks
  // against 0.47 ms for an ordinary pass over a preimage of the shard's size, profiles/r04/chain_scaling_model_1mib.json.)
  hipStream_t side = c->side;
  b3w_chain::Exchange &x = c->x;
  // equal shards (config 4: 1 024 chunks over 8 ranks): no padding, so the collective takes the chunk CVs where they lie and leaves
  // them in global chunk order — no staging copy, no compaction (each a hipMemcpyAsync of its own: 17 of them cost 0.2 ms at 8 ranks)
  const bool even = c->n_chunks % (uint64_t)b3w_comm_size(comm) == 0;
  const uint32_t *d_all = even ? c->d_levels : x.d_cv_all;        // (even: level 0 of the tree IS the receive buffer)
  hipError_t e = hipStreamWaitEvent(side, c->ev_cvs, 0);
  if (e == hipSuccess) e = hipEventRecord(x.ev[0], side);
  if (e == hipSuccess && c->nl && !even) e = hipMemcpyAsync(x.d_cv_pad, c->d_cvs, (uint64_t)c->nl * 32, hipMemcpyDeviceToDevice, side);
  rc = e == hipSuccess ? b3w_comm_allgather(comm, even ? c->d_cvs : x.d_cv_pad, even ? c->d_levels : x.d_cv_gath, x.mx_chunks * 32, side)
                       : hip_fail(ctx, e, "chunk CV staging");
  for (int32_t r = 0; r < b3w_comm_size(comm) && rc == B3W_OK && !even; r++) {        // drop the padding: global chunk order
    uint64_t f = 0; uint32_t k = 0;
    b3w_chain_shard(c->n_chunks, r, b3w_comm_size(comm), &f, &k);
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
  if ((rc = chain_exchange(c, b3w_comm_size(comm))) != B3W_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  b3w_chain::Exchange &x = c->x;
  // wire format per rank: [leaf h_out, mx_leaf rows | parent h_out, mx_par rows], 8 words a row
  HIP_TRY(ctx, hipEventRecord(x.ev[2], st));
  int e = b3w_launch_pack_hout(c->d_pub, 0, c->n_leaf, x.d_h_send, st);
  if (e == 0) e = b3w_launch_pack_hout(c->d_pub, c->n_leaf, c->n_par, x.d_h_send + x.mx_leaf * 8, st);
  if (e) return hip_fail(ctx, (hipError_t)e, "h_out packing");
  const uint64_t block_words = (x.mx_leaf + x.mx_par) * 8;
  if ((rc = b3w_comm_allgather(comm, x.d_h_send, x.d_h_recv, block_words * 4, stream)) != B3W_OK) return rc;
  e = b3w_launch_unpack_hout(x.d_h_recv, block_words, x.mx_leaf * 8, x.d_tab, (uint32_t)b3w_comm_size(comm), x.mx_leaf + x.mx_par, d_leaf_hout,
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
