#!/usr/bin/env python3
"""overlap_cu_partition.py — the commit kernel and the witness kernel on DISJOINT sets of CUs (both streams made with
hipExtStreamCreateWithCUMask), round 6.

The pairings of r05 / r06 (overlap_commit_probe.py) all share every CU between the two kernels, and all give back the sum of the two
times; the pattern — a writer of four store-only waves per CU stops the commitments completely — reads like a wave that needs LOADS being
starved behind the backed-up STORES of the waves beside it on the same CU (the vector-memory path is in order).  If that is the mechanism,
a commit kernel on CUs that carry no writer wave keeps its pace, and the question becomes how few CUs the writer needs.
r04 masked the commit stream alone (B3W_COMMIT_CU_PCT): the witness kernel then still ran on the commit kernel's CUs.

For X in the list: witness kernel on X CUs (mask bits 0 .. X-1: CU ids are dealt to the XCDs round-robin, so a prefix is spread evenly),
commit kernel on the other 256 - X; each alone on its set, then both; HIP events on each stream.
  python tools/ubench/overlap_cu_partition.py [n=65536] [X ...]
"""
import ctypes, importlib, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

m = importlib.import_module("hot-proofs-blake3-circom_amd")
K = importlib.import_module("hot-proofs-blake3-circom_amd.synthetic_key")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
xs = [int(a) for a in sys.argv[2:]] or [256, 192, 160, 128, 96, 64, 48, 32]
dev = torch.device("cuda", 0)
ncu = torch.cuda.get_device_properties(0).multi_processor_count
hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]


def masked_stream(cus):
    """a stream confined to the CUs in `cus` (None: an ordinary stream)"""
    if cus is None:
        return torch.cuda.Stream()
    words = (ncu + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for cu in cus:
        mask[cu // 32] |= 1 << (cu % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), words, mask)
    assert rc == 0, f"hipExtStreamCreateWithCUMask: {rc}"
    return torch.cuda.ExternalStream(s.value)


ctx = m.Context("nova_vesta", 0)
key = m.CommitKey(ctx, "vesta", K.generators("vesta", ctx.witness_size, seed=b"bench"), fold=True)
recs = m.workloads.config3_nova(n)
d_recs = torch.from_numpy(recs.view(np.int32)).to(dev)
bodies = ctx.alloc_bodies(n * ctx.body_bytes)
d_pub = torch.zeros((n, 15), dtype=torch.int32, device=dev)
d_st = torch.zeros(n, dtype=torch.int32, device=dev)
d_st2 = torch.zeros(n, dtype=torch.int32, device=dev)
d_pts = torch.zeros((n, 64), dtype=torch.uint8, device=dev)


def commit(s):
    key.commit_records_device(d_recs.data_ptr(), n, d_pts.data_ptr(), d_st2.data_ptr(), 0, s.cuda_stream)


def witness(s):
    ctx.run_device(d_recs.data_ptr(), n, bodies.ptr, 0, d_pub.data_ptr(), d_st.data_ptr(), s.cuda_stream)


def timed(sa, fa, sb=None, fb=None, reps=5):
    """median ms: (pair wall, fa's own, fb's own); fb is enqueued first, as the chain does"""
    out = []
    for _ in range(reps + 2):
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        e[0].record(sa)
        if fb is not None:
            sb.wait_event(e[0])
            e[3].record(sb); fb(sb); e[4].record(sb)
        e[1].record(sa); fa(sa); e[2].record(sa)
        if fb is not None:
            sa.wait_event(e[4])
        e[5].record(sa)
        torch.cuda.synchronize()
        out.append((e[0].elapsed_time(e[5]), e[1].elapsed_time(e[2]), e[3].elapsed_time(e[4]) if fb is not None else 0.0))
    out = out[2:]
    return tuple(sorted(v[i] for v in out)[len(out) // 2] for i in range(3))


ref = m.Context("nova_vesta", 0)                              # the points of an unmasked run: masks must not change results
d_ref = torch.zeros_like(d_pts)
key.commit_records_device(d_recs.data_ptr(), n, d_ref.data_ptr(), d_st2.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
print(f"n = {n} nova_vesta steps, {ncu} CUs, bodies {bodies.placement}, key window {key.window}")
print(f"{'witness CUs':>11s} {'commit CUs':>10s} | {'witness alone':>13s} {'commit alone':>12s} | {'pair':>8s} {'witness':>8s} {'commit':>8s}   (ms; unmasked pair first)")
for x in xs:
    if x >= ncu:
        sa, sb = masked_stream(None), masked_stream(None)
    else:
        sa, sb = masked_stream(range(0, x)), masked_stream(range(x, ncu))
    wa = timed(sa, witness)[0]
    ca = timed(sb, commit)[0]
    pw, pa, pb = timed(sa, witness, sb, commit)
    assert torch.equal(d_pts, d_ref), "the commitments changed under a CU mask"
    print(f"{min(x, ncu):11d} {ncu - x if x < ncu else ncu:10d} | {wa:13.3f} {ca:12.3f} | {pw:8.3f} {pa:8.3f} {pb:8.3f}", flush=True)
