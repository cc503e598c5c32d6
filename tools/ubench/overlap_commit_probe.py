#!/usr/bin/env python3
"""Why do the commit kernels (vector-ALU bound) and the witness kernel (HBM-write bound) overlap so poorly?  (r05: the pair takes
14.7 ms per 65 536 nova steps side by side against 7.0 + 9.2 one after the other; the bound of a perfect overlap is 9.2.)

Times, for one slice of n nova/Vesta steps: the commitments from the records alone, the witness kernel alone, a pure-store kernel of
the same bytes alone (no LDS, a dozen registers: b3w_store_streams_kernel) — and each writer BESIDE the commitments on a second
stream, with every kernel's own duration from HIP events on its stream.  If the pure-store writer overlaps well and the witness
kernel does not, what stands in the way is residency (LDS / registers), not the memory system.
  [PROBE_COMMIT_PRIO=-1] python tools/ubench/overlap_commit_probe.py [n]
What r05 found (profiles/r05/overlap/): a store-only writer of ONE wave per CU — no LDS, a dozen registers, 6.96 ms alone — next to
the commitments: 10.0 ms for the writer, 14.1 ms for the pair; two waves per CU: the writer keeps its 7.6 ms and the commitments make
no progress while it runs (pair 17.0 = 7.6 + 9.1 + 0.3).  Residency is not what stands in the way; nor are the commit kernel's table
reads (gather_beside_writer.py: an L2-resident table changes nothing): every pairing gives back 0.9 x the sum of the two times.
"""
import ctypes, importlib, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

m = importlib.import_module("hot-proofs-blake3-circom_amd")
K = importlib.import_module("hot-proofs-blake3-circom_amd.synthetic_key")
L = m.lib()
L.b3w_place_store_launch.restype = ctypes.c_int
L.b3w_place_store_launch.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device("cuda", 0)
ctx = m.Context("nova_vesta", 0)
key = m.CommitKey(ctx, "vesta", K.generators("vesta", ctx.witness_size, seed=b"bench"), fold=True)
recs = m.workloads.config3_nova(n)
d_recs = torch.from_numpy(recs.view(np.int32)).to(dev)
bodies = ctx.alloc_bodies(n * ctx.body_bytes)
d_pub = torch.zeros((n, 15), dtype=torch.int32, device=dev)
d_st = torch.zeros(n, dtype=torch.int32, device=dev)
d_st2 = torch.zeros(n, dtype=torch.int32, device=dev)
d_pts = torch.zeros((n, 64), dtype=torch.uint8, device=dev)
prio = int(os.environ.get("PROBE_COMMIT_PRIO", "0"))        # the commit stream's priority (-1 = high)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream(priority=prio)


def commit(s):
    key.commit_records_device(d_recs.data_ptr(), n, d_pts.data_ptr(), d_st2.data_ptr(), 0, s.cuda_stream)


def witness(s):
    ctx.run_device(d_recs.data_ptr(), n, bodies.ptr, 0, d_pub.data_ptr(), d_st.data_ptr(), s.cuda_stream)


def store(shape):
    def f(s):
        assert L.b3w_place_store_launch(bodies.ptr, ctx.body_bytes, n, ctx.body_bytes, shape, s.cuda_stream) == 0
    return f


def timed(fa, fb=None, reps=5):
    """ms of fa alone (fb None), or (wall of the pair, fa's own, fb's own) side by side — fb is enqueued FIRST (the chain enqueues
    the commitments of a batch before its witness kernel), each on its own stream"""
    out = []
    for _ in range(reps + 2):
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        e[0].record(sa)
        sb.wait_event(e[0])
        if fb is not None:
            e[3].record(sb); fb(sb); e[4].record(sb)
        e[1].record(sa); fa(sa); e[2].record(sa)
        if fb is not None:
            sa.wait_event(e[4])
        e[5].record(sa)
        torch.cuda.synchronize()
        out.append((e[0].elapsed_time(e[5]), e[1].elapsed_time(e[2]), e[3].elapsed_time(e[4]) if fb is not None else 0.0))
    out = out[2:]
    return tuple(sorted(x[i] for x in out)[len(out) // 2] for i in range(3))


print(f"n = {n} nova_vesta steps, bodies {bodies.placement}, key window {key.window}, commit stream priority {prio}; the commitments are enqueued first")
print(f"alone   {'commit from records':34s} {timed(commit)[0]:8.3f} ms")
writers = [("witness kernel (8 bodies a wave)", witness), ("pure stores, streams w8 (fills every wave slot)", store(1))]
writers += [(f"pure stores, {k} persistent wave(s) per CU", store(200 + k)) for k in (1, 2, 3, 4, 8)]
# r06: the same stores from a wave that ALLOCATES 8 / 16 / 32 / 64 VGPRs (b3w_store_tiny_kernel): the commit kernel holds 3 x 168 of a SIMD's 512
writers += [(f"tiny writer, {k} wave(s) per CU, {regs} VGPRs allocated", store(base + k)) for k in (1, 2, 4) for base, regs in ((300, 8), (400, 16), (500, 32), (600, 64))]
if os.environ.get("PROBE_ONLY_TINY"):
    writers = [w for w in writers if w[0].startswith(("tiny writer, 1", "witness kernel (8"))]
# (r06 also tried the witness kernel on ONE persistent wave per CU, 8 / 4 / 2 bodies a wave — the shape of the tiny writer, which overlaps best:
# 11.3 / 13.1 / 15.6 ms alone, pairs of 15.8-16.2 ms; profiles/r06/overlap/overlap_one_wave_per_cu.log; those variants are gone again)
for v, what in (("2", "4 bodies a wave"), ("0", "2 bodies a wave"), ("4", "8 bodies a wave, 512 persistent waves")):
    os.environ["B3W_VARIANT"] = v
    ctx2 = m.Context("nova_vesta", 0)
    writers.append((f"witness kernel ({what})", (lambda c: lambda s: c.run_device(d_recs.data_ptr(), n, bodies.ptr, 0, d_pub.data_ptr(), d_st.data_ptr(), s.cuda_stream))(ctx2)))
for name, f in writers:
    alone = timed(f)[0]
    w, a, b = timed(f, commit)
    print(f"beside  {name:48s} pair {w:8.3f} ms   writer {a:8.3f} ms   commit {b:8.3f} ms   (writer alone {alone:.3f} ms)")
