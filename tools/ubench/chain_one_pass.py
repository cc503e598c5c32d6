#!/usr/bin/env python3
"""chain_one_pass.py [ranks] [preimage MiB] [consumer] [steps per ring buffer: 16384] — a few warm-up passes of the chained pass, a pause, then ONE pass: the subject of
a rocprofv3 timeline (tools/pass_timeline.py prints the kernels and copies behind the last pause with their start offsets).
ranks > 1: rank 0's share through the native sharded path with the one-call stand-in all-gather of chain_scaling_model.py.
consumer: none | check | commit | check+commit (the library's own check and the commitments from the records, b3w_chain_commit_overlap auto)."""
import ctypes, importlib, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
m = importlib.import_module("hot-proofs-blake3-circom_amd")
world = int(sys.argv[1]) if len(sys.argv) > 1 else 1
mib = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
consumer = sys.argv[3] if len(sys.argv) > 3 else "none"
nbytes = int(mib * (1 << 20))
dev = torch.device("cuda", 0)
ctx = m.Context("nova_vesta", 0)
host = torch.from_numpy(m.workloads.lcg_preimage(nbytes, seed=1).copy()).pin_memory()
_hip = ctypes.CDLL("libamdhip64.so")
_hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
filled = set()


def standin(d_send, d_recv, nb, stream):
    places = range(world) if (d_recv, nb) not in filled else range(1)
    filled.add((d_recv, nb))
    for r in places:
        assert _hip.hipMemcpyAsync(d_recv + r * nb, d_send, nb, 3, stream) == 0


comm = m.Comm.external(ctx, 0, world, standin) if world > 1 else None
kw = {}
if "check" in consumer:
    kw["check"] = m.R1cs(ctx)
if "commit" in consumer:
    K = importlib.import_module("hot-proofs-blake3-circom_amd.synthetic_key")
    key = m.CommitKey(ctx, "vesta", K.generators("vesta", ctx.witness_size, seed=b"bench"), fold=True)
    n_max = m.lib().b3w_chain_num_chunks(nbytes) * 64 + 64
    kw["commit_records"] = (key, torch.zeros((n_max, 64), dtype=torch.uint8, device=dev))
batch_steps = int(sys.argv[4]) if len(sys.argv) > 4 else 16384
run = lambda: m.chain.fold_witnesses(ctx, host, batch_steps=batch_steps, ring=2, comm=comm, **kw)
for _ in range(4):
    run()
torch.cuda.synchronize()
time.sleep(0.05)
t0 = time.perf_counter()
out = run()
torch.cuda.synchronize()
print(f"one pass: {(time.perf_counter() - t0) * 1e3:.3f} ms host time, {out['n_leaf_steps'] + out['n_parent_steps']} steps, ranks {world}, consumer {consumer}")
