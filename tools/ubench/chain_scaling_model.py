"""chain_scaling_model.py [preimage MiB] — PREDICTED 1 / 2 / 4 / 8-GPU rates of the chained pass (BASELINE config 4: 1 MiB), from what
one GPU can measure: rank 0's share of the pass at every rank count through the NATIVE sharded path (b3w_chain_run_parents_sharded
+ b3w_chain_allgather_hout over a b3w_comm whose all-gather is a stand-in: it copies this rank's block into every rank's place
on the device, so packing, scatter and the bytes are real and the wire is not), plus RCCL's own ncclAllGather of the two messages
measured on a one-rank communicator (a lower bound of its latency: no peer is waited for).  Not a measurement of scaling."""
import ctypes, importlib, json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
m = importlib.import_module("hot-proofs-blake3-circom_amd")
mib = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
nbytes = int(mib * (1 << 20))
dev = torch.device("cuda", 0)
ctx = m.Context("nova_vesta", 0)
host = torch.from_numpy(m.workloads.lcg_preimage(nbytes, seed=1).copy()).pin_memory()
n_chunks = m.lib().b3w_chain_num_chunks(nbytes)
total_steps = m.lib().b3w_chain_num_leaf_steps(nbytes) + m.lib().b3w_chain_parent_row(n_chunks, n_chunks)


_hip = ctypes.CDLL("libamdhip64.so")
_hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
_hip.hipMemcpyAsync.restype = ctypes.c_int


def standin(world):
    """this rank's block into its own place of the gathered buffer, device to device on the exchange's stream — ONE copy call, as the
    real collective is one call; the other ranks' places keep what the first pass put there (every place filled once: the bytes the
    tree, the parent plan and the scatter kernel then read are real).  r04's stand-in made `world` copy calls per exchange: at 8 ranks
    0.19 ms of the side stream were the stand-in's own launches, as long as the rank's leaf witness kernel beside it."""
    filled = set()

    def fn(d_send, d_recv, nbytes_per_rank, stream):
        key = (d_recv, nbytes_per_rank)
        places = range(world) if key not in filled else range(1)
        filled.add(key)
        for r in places:
            if _hip.hipMemcpyAsync(d_recv + r * nbytes_per_rank, d_send, nbytes_per_rank, 3, stream) != 0:
                raise RuntimeError("hipMemcpyAsync")
    return fn


rows = []
for world in (1, 2, 4, 8):
    comm = m.Comm.external(ctx, 0, world, standin(world)) if world > 1 else None
    run = lambda: m.chain.fold_witnesses(ctx, host, batch_steps=16384, ring=2, comm=comm)
    def timed(fn, reps=20):
        """-> (ms per pass with `reps` passes queued back to back, median ms of one pass waited for, the last pass's result)"""
        for _ in range(3):
            o = fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            o = fn()
        torch.cuda.synchronize()
        queued = (time.perf_counter() - t0) / reps * 1e3
        one = []
        for _ in range(reps):
            t0 = time.perf_counter()
            o = fn()
            torch.cuda.synchronize()
            one.append((time.perf_counter() - t0) * 1e3)
        return queued, sorted(one)[len(one) // 2], o
    pass_ms, pass_one_ms, out = timed(run)
    ex = m.chain.exchange_ms(out)
    steps = out["n_leaf_steps"] + out["n_parent_steps"]
    # beside it: an ordinary one-rank pass over a preimage of the shard's size (what the sharded path costs beyond its share of the steps)
    small = host[: nbytes // world]
    small_ms, small_one_ms, small_out = timed(lambda: m.chain.fold_witnesses(ctx, small, batch_steps=16384, ring=2))
    rows.append(dict(ranks=world, rank0_steps=steps, rank0_pass_ms=round(pass_ms, 3), rank0_single_pass_median_ms=round(pass_one_ms, 3),
                     standin_exchange_ms=[round(ex[0], 3), round(ex[1], 3)],
                     one_rank_pass_of_a_preimage_of_the_shards_size_ms=round(small_ms, 3), its_single_pass_median_ms=round(small_one_ms, 3),
                     its_steps=small_out["n_leaf_steps"] + small_out["n_parent_steps"]))
    if comm is not None:
        comm.close()
    ctx.trim()

# RCCL's own call, one rank: the two messages of the 8-rank exchange (chunk CVs: n_chunks / 8 x 32 B; h_out: (steps / 8) x 32 B)
rccl = {}
try:
    c1 = m.Comm(ctx, m.Comm.unique_id(), 0, 1)
    for name, nb in (("chunk_cvs", max(n_chunks // 8, 1) * 32), ("h_out", max(total_steps // 8, 1) * 32)):
        a = torch.zeros(nb, dtype=torch.uint8, device=dev); b = torch.zeros(nb, dtype=torch.uint8, device=dev)
        s = torch.cuda.current_stream().cuda_stream
        for _ in range(5): c1.allgather(a.data_ptr(), b.data_ptr(), nb, s)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): c1.allgather(a.data_ptr(), b.data_ptr(), nb, s)
        e1.record(); torch.cuda.synchronize()
        rccl[name] = {"bytes_per_rank": nb, "one_rank_ncclAllGather_us": round(e0.elapsed_time(e1) / 50 * 1e3, 2)}
    c1.close()
except Exception as e:                                       # (no librccl on this box)
    rccl = {"error": str(e)}
lat = sum(v["one_rank_ncclAllGather_us"] for v in rccl.values()) * 1e-3 if "error" not in rccl else 0.0
for r in rows:
    for label, extra in (("with_one_rank_rccl_latency", lat), ("with_50us_per_collective", 0.1)):
        t = r["rank0_single_pass_median_ms"] + (extra if r["ranks"] > 1 else 0.0)   # (a config-4 fold IS one pass: its latency counts)
        r["predicted_M_steps_per_s_" + label] = round(total_steps / t / 1e3, 3)
print(json.dumps({"what": "PREDICTED, not measured: rank 0's share through the native sharded path with a stand-in all-gather on one GPU",
                  "preimage_mib": mib, "total_steps": int(total_steps), "rccl_one_rank": rccl, "rows": rows}, indent=1))
