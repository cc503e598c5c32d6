// b3w_wtns.cpp — C-ABI part 3: the streaming .wtns writer (b3w_batch_write_wtns[_ex]).
#include "b3w_internal.h"

extern "C" {

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
}  // extern "C"
