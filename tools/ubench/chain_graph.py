#!/usr/bin/env python3
"""chain_graph.py [ranks] [preimage MiB] — passes of the chained pass queued back to back: launched call by call, and replayed from a
hipGraph captured once (tests/test_gpu_graph_capture.py::test_a_whole_chained_pass_replays_from_a_graph).  A 1 MiB pass is ≈ 25 launches
and a dozen event operations; queued eagerly the host falls behind the device (chain_scaling_model.py: 1.1 ms per queued pass of rank
0's share at 8 ranks against 0.50 ms for one pass waited for)."""
import ctypes, importlib, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
m = importlib.import_module("hot-proofs-blake3-circom_amd")
world = int(sys.argv[1]) if len(sys.argv) > 1 else 1
mib = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
nbytes = int(mib * (1 << 20))
ctx = m.Context("nova_vesta", 0)
host = torch.from_numpy(m.workloads.lcg_preimage(nbytes, seed=1).copy()).pin_memory()
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]


def standin(d_send, d_recv, nb, stream):
    assert hip.hipMemcpyAsync(d_recv, d_send, nb, 3, stream) == 0


comm = m.Comm.external(ctx, 0, world, standin) if world > 1 else None
fold = lambda: m.chain.fold_witnesses(ctx, host, batch_steps=16384, ring=2, comm=comm)
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    for _ in range(3):
        out = fold()
    side.synchronize()
    steps = out["n_leaf_steps"] + out["n_parent_steps"]
    reps = 50
    t0 = time.perf_counter()
    for _ in range(reps):
        fold()
    side.synchronize()
    eager = (time.perf_counter() - t0) / reps * 1e3
g = torch.cuda.CUDAGraph()
with m.graph_capture(g, stream=side):
    fold()
g.replay(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    g.replay()
torch.cuda.synchronize()
graph = (time.perf_counter() - t0) / reps * 1e3
one = []
for _ in range(20):
    t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); one.append((time.perf_counter() - t0) * 1e3)
print(f"ranks {world}, {mib:g} MiB, rank 0's {steps} steps per pass: queued call by call {eager:.3f} ms per pass = {steps / eager / 1e3:.2f} M steps/s; "
      f"replayed from a graph {graph:.3f} ms = {steps / graph / 1e3:.2f} M steps/s; one replay waited for: {sorted(one)[10]:.3f} ms")
