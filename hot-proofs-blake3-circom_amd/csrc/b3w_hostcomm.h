// b3w_hostcomm.h — all-gather between the processes of one host through a POSIX shared-memory segment (b3w_hostcomm.cpp).
// The transport behind b3w_comm_create_host: ranks that share a GPU, or hosts without RCCL.  No HIP in here: host pointers only.
#pragma once
#include <stddef.h>
#include <stdint.h>

struct B3wHostComm;

// Every rank of the job passes the same `name` (a POSIX shared-memory name, "/..." — unique to the job: rank 0 removes a
// segment of that name it finds and creates a fresh one, the others wait for it), `nranks` and `slot_bytes` (bytes per rank
// the segment can carry at a time; larger messages go through in pieces).  Returns 0, or -1 with a message in err.
// Every wait (for the segment, at a barrier) gives up after timeout_s seconds and poisons the segment: all ranks then fail.
int b3w_hostcomm_open(const char *name, int rank, int nranks, uint64_t slot_bytes, double timeout_s, B3wHostComm **out, char *err,
                      size_t errlen);
// recv[r * bytes_per_rank ..) = rank r's send[0 .. bytes_per_rank), on every rank.  Blocking; all ranks must call it with the
// same bytes_per_rank.  0, or -1 with a message in err.
int b3w_hostcomm_allgather(B3wHostComm *c, const void *send, void *recv, uint64_t bytes_per_rank, char *err, size_t errlen);
void b3w_hostcomm_close(B3wHostComm *c);
