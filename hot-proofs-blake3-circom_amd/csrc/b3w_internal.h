// b3w_internal.h — what the files behind include/b3wit.h share: the context and batch objects, the error and device-guard
// plumbing, and the few functions one object's file needs of another's.  Not installed; nothing here is part of the C-ABI.
#pragma once
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
#include "b3w_capture.h"
#include "b3w_kernels.h"
#include "b3w_r1cs_host.h"


struct b3w_layout_run { char kind; uint32_t slot, atom, bit0, len; };

struct InputSignal { const char *name; uint32_t count; uint32_t rec_off; uint64_t hash; };

struct CircuitDesc {
  int kind;
  uint32_t nwit, nin, npub;
  const uint64_t *prime;
  const b3w_layout_run *runs;
  uint32_t nruns;
  uint32_t lds_words;
};
inline constexpr uint64_t P_BN254[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
inline constexpr uint64_t P_VESTA[4] = {0x8c46eb2100000001ull, 0x224698fc0994a8ddull, 0x0ull, 0x4000000000000000ull};
constexpr uint32_t B3W_INV_TABLE_N = 2048;      // small inverses the nova kernels (and the commit keys' tables) look IsZero arguments up in

struct b3w_ctx {
  int circuit = -1;
  CircuitDesc desc{};
  int device = -1;
  int variant = 0;
  bool variant_auto = true;           // no B3W_VARIANT and no autotune yet: launch shape chosen by batch size
  bool fill_ok = false;               // the slot table can be said in 16 bits a slot (b3w_kernels.hip fill_entry16): the REGIONFILL variant applies
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

inline int32_t hip_fail(b3w_ctx *ctx, hipError_t e, const char *what) {
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

// ---- one object's file to another's
// b3w_ctx.cpp: the slot table of a circuit (slot -> LDS word, shift, mode), as the witness kernels and the commit keys read it
bool b3w_int_build_slot_table(const CircuitDesc &c, std::vector<uint32_t> &table, std::string &err);
int b3w_int_default_variant(const b3w_ctx *ctx, uint32_t n, const uint8_t *d_bodies, uint64_t pitch);    // the default launch shape for this batch into this buffer
b3w_ctx *b3w_int_key_ctx(const b3w_commit_key *key);
// b3w_commit_api.cpp: b3w_commit_records_device with the normalisation put off — d_sums_out (B3W_COMMIT_SUM_WORDS words per record) takes
// the projective sums and d_points is not written; b3w_int_commit_normalize makes points of any number of them at once (the chained pass:
// once per run call instead of once per batch, b3w_kernels.h at b3w_launch_commit_normalize)
// trace_done (an event the caller owns; null: both launches on `stream`): the TRACE launch goes to trace_stream — ANOTHER stream than the commit
// kernel's, and the null stream is a stream — and `stream` waits for the event (n at most one chunk of 32 768) — chain_run_steps: the images are 115 MB of stores that crawl beside a
// witness kernel but take 40 us in front of it
int32_t b3w_int_commit_records(b3w_ctx *ctx, const b3w_commit_key *key, const uint32_t *d_records, uint32_t n, uint8_t *d_points, uint32_t *d_public,
                               int32_t *d_status, void *stream, uint32_t *d_sums_out, void *trace_stream = nullptr, hipEvent_t trace_done = nullptr);
int32_t b3w_int_commit_normalize(b3w_ctx *ctx, const b3w_commit_key *key, const uint32_t *d_sums, uint64_t n, uint8_t *d_points, void *stream);
b3w_ctx *b3w_int_r1cs_ctx(const b3w_r1cs *r1cs);
constexpr uint64_t B3W_RING_SPARE_CAP = 26ull << 30;    // ring buffers a context keeps between chains: two 16 384-step nova buffers
