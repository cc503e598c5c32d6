// b3w_hostcomm.cpp — all-gather between the processes of one host through one POSIX shared-memory segment.
//
// What it is for: b3w_comm_create_host (include/b3wit.h).  The sharded chained pass has two exchanges — the chunk chaining
// values and every step's h_out (the z_{i+1} the reference's fold feeds back, rust_fold/src/blake3_circuit.rs:111-123,
// rust_fold/src/main.rs:166-179).  Over xGMI they are ncclAllGather calls; this file carries the same calls between
// processes that share one GPU (or have no RCCL), so that the rank > 0 paths of the exchange run on a one-GPU box.
//
// Segment: one header page, then nranks slots of slot_bytes.  An all-gather of B bytes per rank goes through in pieces of at
// most slot_bytes: every rank copies its piece into its slot, barrier, every rank copies all slots out, barrier.  The barrier
// is a counter + generation word in the header; every wait has a deadline, and a rank that runs into it poisons the segment so
// that its peers fail too instead of waiting for it for good.
#include "b3w_hostcomm.h"

#include <errno.h>
#include <fcntl.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <new>
#include <string>

namespace {
constexpr uint32_t MAGIC = 0x42335748u;          // "B3WH"
constexpr size_t HEADER_BYTES = 4096;

struct Header {
  std::atomic<uint32_t> magic;                   // set last by rank 0
  uint32_t nranks;
  uint64_t slot_bytes;
  int64_t owner_pid;                             // rank 0's process: a segment whose owner is gone is a stale one
  std::atomic<uint32_t> arrived, generation, poisoned;
};
static_assert(sizeof(Header) <= HEADER_BYTES, "header page");
static_assert(std::atomic<uint32_t>::is_always_lock_free, "cross-process atomics need lock-free words");

double now_s() {
  timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

void set_err(char *err, size_t errlen, const std::string &s) {
  if (err && errlen) snprintf(err, errlen, "%s", s.c_str());
}

void nap(uint32_t spins) {
  if (spins < 2000) {
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
    return;
  }
  timespec t = {0, spins < 20000 ? 20000 : 200000};          // 20 us, later 200 us: ranks may outnumber the cores
  nanosleep(&t, nullptr);
}
}  // namespace

struct B3wHostComm {
  std::string name;
  int rank = 0, nranks = 1;
  uint64_t slot_bytes = 0;
  double timeout_s = 120;
  size_t map_bytes = 0;
  uint8_t *base = nullptr;
  bool linked = false;                           // the name still exists (rank 0 removes it once everybody is attached)
  Header *hdr() const { return reinterpret_cast<Header *>(base); }
  uint8_t *slot(int r) const { return base + HEADER_BYTES + (size_t)r * slot_bytes; }
};

namespace {
int barrier(B3wHostComm *c, char *err, size_t errlen) {
  Header *h = c->hdr();
  if (h->poisoned.load(std::memory_order_acquire)) { set_err(err, errlen, "host communicator: a peer gave up (timeout or error)"); return -1; }
  const uint32_t g = h->generation.load(std::memory_order_acquire);
  if (h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)c->nranks) {
    h->arrived.store(0, std::memory_order_relaxed);
    h->generation.store(g + 1, std::memory_order_release);
    return 0;
  }
  const double t_end = now_s() + c->timeout_s;
  for (uint32_t spins = 0; h->generation.load(std::memory_order_acquire) == g; ++spins) {
    if (h->poisoned.load(std::memory_order_acquire)) { set_err(err, errlen, "host communicator: a peer gave up (timeout or error)"); return -1; }
    if ((spins & 1023) == 1023 && now_s() > t_end) {
      h->poisoned.store(1, std::memory_order_release);
      char msg[160];
      snprintf(msg, sizeof msg, "host communicator: rank %d waited %.0f s at a barrier for its %d peers", c->rank, c->timeout_s, c->nranks - 1);
      set_err(err, errlen, msg);
      return -1;
    }
    nap(spins);
  }
  return 0;
}
}  // namespace

int b3w_hostcomm_open(const char *name, int rank, int nranks, uint64_t slot_bytes, double timeout_s, B3wHostComm **out, char *err,
                      size_t errlen) {
  if (!out) return -1;
  *out = nullptr;
  if (!name || name[0] != '/' || strchr(name + 1, '/') || strlen(name) > 200 || nranks < 1 || rank < 0 || rank >= nranks || !slot_bytes) {
    set_err(err, errlen, "host communicator: name must look like \"/job-unique-name\", 0 <= rank < nranks, slot_bytes > 0");
    return -1;
  }
  slot_bytes = (slot_bytes + 63) & ~(uint64_t)63;
  B3wHostComm *c = new (std::nothrow) B3wHostComm;
  if (!c) { set_err(err, errlen, "host communicator: out of memory"); return -1; }
  c->name = name; c->rank = rank; c->nranks = nranks; c->slot_bytes = slot_bytes; c->timeout_s = timeout_s > 0 ? timeout_s : 120;
  c->map_bytes = HEADER_BYTES + (size_t)nranks * slot_bytes;
  const double t_end = now_s() + c->timeout_s;
  if (rank == 0) {
    (void)shm_unlink(name);                                   // a segment a dead job left under this name
    const int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) {
      set_err(err, errlen, std::string("host communicator: cannot create ") + name + ": " + strerror(errno));
      if (fd >= 0) { close(fd); (void)shm_unlink(name); }
      delete c;
      return -1;
    }
    void *p = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) {
      set_err(err, errlen, std::string("host communicator: mmap: ") + strerror(errno));
      (void)shm_unlink(name);
      delete c;
      return -1;
    }
    c->base = static_cast<uint8_t *>(p);
    c->linked = true;
    Header *h = new (c->base) Header;                         // (a fresh segment is zero-filled)
    h->nranks = (uint32_t)nranks; h->slot_bytes = slot_bytes; h->owner_pid = (int64_t)getpid();
    h->arrived.store(0); h->generation.store(0); h->poisoned.store(0);
    h->magic.store(MAGIC, std::memory_order_release);
  } else {
    for (uint32_t spins = 0;; ++spins) {
      const int fd = shm_open(name, O_RDWR, 0600);
      if (fd >= 0) {
        struct stat st;
        if (fstat(fd, &st) == 0 && (size_t)st.st_size >= c->map_bytes) {
          void *p = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
          if (p != MAP_FAILED) {
            Header *h = static_cast<Header *>(p);
            const bool ready = h->magic.load(std::memory_order_acquire) == MAGIC;
            // ours: made for this geometry by a process that is still there (a stale segment is about to be replaced by rank 0)
            if (ready && h->nranks == (uint32_t)nranks && h->slot_bytes == slot_bytes && (kill((pid_t)h->owner_pid, 0) == 0 || errno == EPERM) &&
                !h->poisoned.load(std::memory_order_acquire)) {
              c->base = static_cast<uint8_t *>(p);
              close(fd);
              break;
            }
            munmap(p, c->map_bytes);
          }
        }
        close(fd);
      }
      if (now_s() > t_end) {
        set_err(err, errlen, std::string("host communicator: rank 0 never created ") + name);
        delete c;
        return -1;
      }
      timespec t = {0, spins < 100 ? 100000 : 2000000};
      nanosleep(&t, nullptr);
    }
  }
  if (barrier(c, err, errlen) != 0) { b3w_hostcomm_close(c); return -1; }   // everybody holds a mapping
  if (rank == 0) { (void)shm_unlink(name); c->linked = false; }             // ... so the name can go: nothing is left behind by a crash
  // second phase: nobody returns while the name still exists.  (With one barrier a rank could leave, close, and re-open a communicator
  // of the same name — a retry loop — while rank 0 had not unlinked yet: it attached to the OLD segment, alive owner and right
  // geometry, and both sides waited for each other until their timeouts.)
  if (barrier(c, err, errlen) != 0) { b3w_hostcomm_close(c); return -1; }
  *out = c;
  return 0;
}

int b3w_hostcomm_allgather(B3wHostComm *c, const void *send, void *recv, uint64_t bytes_per_rank, char *err, size_t errlen) {
  if (!c || !send || !recv) { set_err(err, errlen, "host communicator: bad argument"); return -1; }
  const uint8_t *s = static_cast<const uint8_t *>(send);
  uint8_t *d = static_cast<uint8_t *>(recv);
  for (uint64_t off = 0; off < bytes_per_rank; off += c->slot_bytes) {
    const uint64_t k = bytes_per_rank - off < c->slot_bytes ? bytes_per_rank - off : c->slot_bytes;
    memcpy(c->slot(c->rank), s + off, k);
    if (barrier(c, err, errlen) != 0) return -1;              // every slot is written
    for (int r = 0; r < c->nranks; r++) memcpy(d + (uint64_t)r * bytes_per_rank + off, c->slot(r), k);
    if (barrier(c, err, errlen) != 0) return -1;              // every slot has been read: it may be overwritten
  }
  return 0;
}

void b3w_hostcomm_close(B3wHostComm *c) {
  if (!c) return;
  if (c->base) munmap(c->base, c->map_bytes);
  if (c->linked) (void)shm_unlink(c->name.c_str());
  delete c;
}
