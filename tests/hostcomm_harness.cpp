// hostcomm_harness.cpp — csrc/b3w_hostcomm.cpp (the shared-memory all-gather behind b3w_comm_create_host) between forked
// processes, no GPU: every rank sends a pattern that names (rank, round, offset) and checks every byte it receives.
//   harness run <nranks> <slot_bytes>      all ranks present: message sizes below, at and above one slot
//   harness missing <nranks>               the last rank never comes: every other rank must fail within the timeout
//   harness stale <nranks>                 a dead job's segment lies under the name: the job replaces it
//   harness reopen <nranks>                every rank opens, gathers, closes and re-opens under the SAME name, five times in a row
//                                          (a rank must not find the previous round's segment still linked)
//   harness threads <nranks>               the ranks are THREADS of one process (how eight ranks are rehearsed on a box that admits
//                                          six processes to its GPU: tests/test_gpu_native_exchange.py)
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

#include <string>
#include <thread>
#include <vector>

#include "b3w_hostcomm.h"

static uint8_t pat(int rank, int round, uint64_t i) { return (uint8_t)(rank * 131 + round * 29 + i * 7 + (i >> 8)); }

static int rank_main(const std::string &name, int rank, int nranks, uint64_t slot, double timeout, bool expect_fail);

static int reopen_main(const std::string &name, int rank, int nranks, uint64_t slot) {
  for (int round = 0; round < 5; round++) {
    if (rank == nranks - 1 && round) usleep(20000 * round);  // the ranks do not come back at the same time
    const int rc = rank_main(name, rank, nranks, slot, 10.0, false);
    if (rc) { fprintf(stderr, "rank %d: round %d of re-opening failed\n", rank, round); return rc; }
  }
  return 0;
}

static int rank_main(const std::string &name, int rank, int nranks, uint64_t slot, double timeout, bool expect_fail) {
  char err[256] = "";
  B3wHostComm *c = nullptr;
  const int rc = b3w_hostcomm_open(name.c_str(), rank, nranks, slot, timeout, &c, err, sizeof err);
  if (expect_fail) {
    if (rc == 0) { fprintf(stderr, "rank %d: open succeeded without all peers\n", rank); return 1; }
    printf("rank %d failed as it should: %s\n", rank, err);
    return 0;
  }
  if (rc != 0) { fprintf(stderr, "rank %d: open: %s\n", rank, err); return 1; }
  const uint64_t sizes[] = {1, 32, 4097, slot - 1, slot, slot + 1, 3 * slot + 5, 8};
  int round = 0;
  for (uint64_t bytes : sizes) {
    if (!bytes) continue;
    std::vector<uint8_t> send(bytes), recv(bytes * nranks, 0xEE);
    for (uint64_t i = 0; i < bytes; i++) send[i] = pat(rank, round, i);
    if (b3w_hostcomm_allgather(c, send.data(), recv.data(), bytes, err, sizeof err) != 0) { fprintf(stderr, "rank %d: allgather: %s\n", rank, err); return 1; }
    for (int r = 0; r < nranks; r++)
      for (uint64_t i = 0; i < bytes; i++)
        if (recv[(uint64_t)r * bytes + i] != pat(r, round, i)) {
          fprintf(stderr, "rank %d round %d: byte %llu of rank %d's block is wrong\n", rank, round, (unsigned long long)i, r);
          return 1;
        }
    round++;
  }
  b3w_hostcomm_close(c);
  return 0;
}

int main(int argc, char **argv) {
  if (argc < 3) { fprintf(stderr, "usage: %s run|missing|stale <nranks> [slot_bytes]\n", argv[0]); return 2; }
  const std::string mode = argv[1];
  const int nranks = atoi(argv[2]);
  const uint64_t slot = argc > 3 ? strtoull(argv[3], nullptr, 10) : 4096;
  const std::string name = "/b3w_hostcomm_test_" + std::to_string((long)getpid());
  if (mode == "stale") {                                     // what a job that died before its first barrier leaves behind
    const int fd = shm_open(name.c_str(), O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, 4096 + (off_t)nranks * (off_t)((slot + 63) & ~63ull)) != 0) { perror("stale segment"); return 1; }
    void *p = mmap(nullptr, 4096, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    const uint32_t junk[8] = {0x42335748u, (uint32_t)nranks, (uint32_t)((slot + 63) & ~63ull), 0, 0x7fffff00u /* a pid nobody has */, 0, 5, 9};
    memcpy(p, junk, sizeof junk);
    munmap(p, 4096);
    close(fd);
  }
  if (mode == "threads") {
    std::vector<std::thread> th;
    std::vector<int> rcs(nranks, -1);
    for (int r = 0; r < nranks; r++) th.emplace_back([&, r] { rcs[r] = reopen_main(name, r, nranks, slot); });
    for (auto &t : th) t.join();
    int bad = 0;
    for (int rc : rcs) bad += rc != 0;
    if (shm_open(name.c_str(), O_RDWR, 0600) >= 0) { fprintf(stderr, "the segment's name was left behind\n"); shm_unlink(name.c_str()); bad++; }
    printf("%s: %d ranks, %d failed\n", mode.c_str(), nranks, bad);
    return bad ? 1 : 0;
  }
  const bool missing = mode == "missing";
  std::vector<pid_t> kids;
  for (int r = 0; r < nranks - (missing ? 1 : 0); r++) {
    const pid_t pid = fork();
    if (pid == 0) {
      if (mode == "stale" && r == 0) usleep(300000);         // the others look for the segment first and find the stale one
      const int rc = mode == "reopen" ? reopen_main(name, r, nranks, slot) : rank_main(name, r, nranks, slot, missing ? 2.0 : 30.0, missing);
      fflush(stdout);
      _exit(rc);
    }
    kids.push_back(pid);
  }
  int bad = 0;
  for (pid_t k : kids) {
    int st = 0;
    waitpid(k, &st, 0);
    if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) bad++;
  }
  if (shm_open(name.c_str(), O_RDWR, 0600) >= 0) { fprintf(stderr, "the segment's name was left behind\n"); shm_unlink(name.c_str()); bad++; }
  printf("%s: %d ranks, %d failed\n", mode.c_str(), nranks, bad);
  return bad ? 1 : 0;
}
