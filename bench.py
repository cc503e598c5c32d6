#!/usr/bin/env python3
"""bench.py — BLAKE3-compression witnesses/s on MI355X (BASELINE.json metric, config 2).

A step = one pass of the hot path over one batch: 4096 independent blake3_compression witnesses
(BN254) per GPU, inputs already resident in HBM, witness bodies written to HBM (3.16 GB per
step and GPU).  With N > 1 GPUs every rank runs its own 4096 instances (weak scaling; instance
ids rank*4096 ...) and the ranks all-gather the per-witness public outputs over RCCL.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (see the driver contract) with `roofline` (HBM-write bound
kernel: algorithmic bytes / HIP-event kernel time) and `cpu_baseline` (the C oracle timed on
one host core on a bounded sample of the same workload).
"""
import argparse, importlib, json, os, sys, time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (about 6.3 TB/s achievable)
BYTES_PER_WITNESS = {"compression": 770976 + 112, "nova_bn254": 745312 + 128, "nova_vesta": 745312 + 128,
                     "nova_bn254_o1": 787648 + 128}   # body written + record read


def cpu_baseline(circuit, recs, budget_s):
    """Time the oracle (oracle/libb3w_oracle.so, the CPU restatement = "port") on one core over a
    bounded sample of the same records.  Checker code, used here only as the reported baseline."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import b3w_testlib as T
    import numpy as np
    sample = recs[:256]
    T.oracle_batch_u32(circuit, sample)                       # warm-up: faults the output buffer in
    done, t0 = 0, time.perf_counter()
    while True:
        bad, _ = T.oracle_batch_u32(circuit, sample)
        assert bad == 0
        done += sample.shape[0]
        dt = time.perf_counter() - t0
        if dt >= budget_s:
            break
    return {"value": done / dt, "unit": "witnesses/s", "cores": 1, "kind": "port",
            "sample": f"{done} witnesses ({done // sample.shape[0]} passes over the first {sample.shape[0]} "
                      f"records of the workload) in {dt:.1f} s, C oracle, 1 thread"}


def bench_chain(args, m, torch, dist, dev, world, rank, local_rank):
    """Configs 4/5: a step = one pass over the whole preimage (plan + all leaf and parent step witnesses of this
    rank's chunk range, bodies streamed through a ring of batch buffers).  Strong scaling: the preimage is fixed."""
    import numpy as np
    circuit = args.circuit if args.circuit != "compression" else "nova_vesta"
    ctx = m.Context(circuit, local_rank)
    nbytes = int(args.preimage_mib * (1 << 20))
    lcg_words = (np.arange(nbytes // 4 + 1, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(12345)) & np.uint64(0xFFFFFFFF)
    host = torch.from_numpy(lcg_words.astype(np.uint32).view(np.uint8)[:nbytes].copy()).pin_memory()
    consumer, key, commit_only = None, None, None
    if args.consumer != "none":                      # what the folding prover does with each step witness
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import ec_ref as E                             # plain-integer curve arithmetic: here only to make valid generators
        curve = "vesta" if "vesta" in circuit else "bn254_g1"
        key = m.CommitKey(ctx, curve, E.points_to_bytes(E.random_points(curve, ctx.witness_size, seed=b"bench")))
        n_max = m.lib().b3w_chain_num_chunks(nbytes) * 64 + 64
        d_pts = torch.zeros((n_max, 64), dtype=torch.uint8, device=dev)
        d_st = torch.zeros(n_max, dtype=torch.int32, device=dev)

        def consumer(view, first, k):
            key.commit_device(view.data_ptr(), k, view.stride(0), d_pts.data_ptr() + 64 * first, d_st.data_ptr() + 4 * first,
                              torch.cuda.current_stream().cuda_stream)
        if args.consumer == "commit-only":
            consumer, commit_only = None, (key, d_pts)
    run = lambda: m.chain.fold_witnesses(ctx, host, batch_steps=args.batch if args.batch != 4096 else 16384, ring=2, consumer=consumer,
                                         commit_only=commit_only)
    for _ in range(max(3, args.warmup)):        # the first passes pay the allocator (24 GB ring, record buffers)
        out = run()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = run()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    assert int(out["status"].abs().sum().item()) == 0
    local_steps = out["n_leaf_steps"] + out["n_parent_steps"]
    if key is not None:
        assert int(d_st[:local_steps].abs().sum().item()) == 0 and int(d_pts[:local_steps].max(dim=1).values.min().item()) > 0
    t = torch.tensor([elapsed, float(local_steps)], dtype=torch.float64, device=dev)
    if world > 1:
        tm = t.clone(); dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        ts = t.clone(); dist.all_reduce(ts, op=dist.ReduceOp.SUM)
        elapsed, total_steps = tm[0].item(), ts[1].item()
    else:
        total_steps = float(local_steps)
    if rank == 0:
        per = BYTES_PER_WITNESS[circuit] if commit_only is None else 128      # commit-only reads the 128-byte step records
        print(json.dumps({
            "metric": "BLAKE3-compression witnesses/sec", "value": total_steps * args.steps / elapsed, "unit": "witnesses/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"chain: {args.preimage_mib} MiB preimage -> {int(total_steps)} nova steps ({circuit}), "
                                   "planner + witness kernels, bodies through a 2-deep ring, H2D overlapped",
                       "circuit": circuit, "n_chunks": out["n_chunks"], "path_len": out["path_len"],
                       "placement": out.get("placement"),
                       "consumer": "none" if key is None else f"Pedersen commitment of every step witness on the device ({key.window}-bit windows)"
                                   + (", from the step records: no bodies written" if commit_only is not None else "")},
            "roofline": {"bound": "hbm", "achieved": total_steps * args.steps * per / elapsed / 1e9 / world, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": total_steps * args.steps * per / elapsed / 1e9 / world / HBM_PEAK_GBS,
                         "traffic": None, "note": "end-to-end per-GPU rate incl. planner, H2D and launch gaps"},
        }), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4096, help="witnesses per GPU per step")
    ap.add_argument("--circuit", default="compression")
    ap.add_argument("--variant", type=int, default=None, help="kernel tuning variant (B3W_VARIANT)")
    ap.add_argument("--pitch", type=int, default=0, help="body pitch in bytes (0 = contiguous bodies)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--placement", default="mixed", choices=["mixed", "plain"],
                    help="body buffer placement: mixed = b3w_bodies_alloc's two-class buffer (default), plain = hipMalloc")
    ap.add_argument("--workload", default="batch", choices=["batch", "chain"],
                    help="batch = BASELINE config 2/3 (default, the headline metric); chain = configs 4/5: "
                         "preimage -> planner -> nova step witnesses, streamed through a ring of buffers")
    ap.add_argument("--preimage-mib", type=float, default=1.0, help="chain workload: preimage size (1 = config 4, 1024 = config 5)")
    ap.add_argument("--consumer", default="none", choices=["none", "commit", "commit-only"],
                    help="chain workload: what reads each batch of step witnesses while it sits in the ring "
                         "(commit = Pedersen commitments on the circuit's curve, synthetic generators; commit-only = the same "
                         "commitments straight from the step records, no bodies written)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product has no CPU path")
    ndev = torch.cuda.device_count()
    if local_rank >= ndev and os.environ.get("B3W_DIST_BACKEND", "nccl") != "nccl":
        local_rank = local_rank % ndev                      # dry run: several ranks share one GPU
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        backend = os.environ.get("B3W_DIST_BACKEND", "nccl")      # "nccl" = RCCL over xGMI; "gloo" only for dry runs
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    if args.variant is not None:
        os.environ["B3W_VARIANT"] = str(args.variant)
    m = importlib.import_module("hot-proofs-blake3-circom_amd")
    W = m.workloads
    if args.workload == "chain":
        return bench_chain(args, m, torch, dist, dev, world, rank, local_rank)
    circuit, n = args.circuit, args.batch
    ctx = m.Context(circuit, local_rank)
    recs = W.config2_compression(n, first=rank * n) if circuit == "compression" else W.config3_nova(n, first=rank * n)
    pitch = args.pitch or ctx.body_bytes
    d_recs = torch.from_numpy(recs.view(np.int32)).to(dev)
    npub = ctx.public_words
    d_pub = torch.zeros((n, npub), dtype=torch.int32, device=dev)
    d_status = torch.zeros((n,), dtype=torch.int32, device=dev)
    sharding = importlib.import_module("hot-proofs-blake3-circom_amd.sharding")
    stream = torch.cuda.current_stream()

    # the fold's exchange step (N > 1): all-gather of the per-step public outputs (h_out ...), pipelined — the gather
    # of step i overlaps the kernel of step i+1 on RCCL's own stream (sharding.PublicExchange)
    ex = sharding.PublicExchange(n, npub, dev)

    def step():
        pub = ex.next_buffer()
        ctx.run_device(d_recs.data_ptr(), n, d_bodies.data_ptr(), pitch, pub.data_ptr(), d_status.data_ptr(),
                       stream.cuda_stream)
        ex.post()

    # Untimed set-up.  The body buffer comes from the library's placement allocator (b3w_bodies_alloc: its 256 MiB
    # pieces alternate between two classes of HBM, DESIGN.md "Placement"), then the faster of the two bit-identical
    # kernel paths is picked on that buffer.
    if args.placement == "plain":
        os.environ["B3W_PLACEMENT"] = "plain"
    bodies = ctx.alloc_bodies(n * pitch)
    for attempt in range(4):                                # a box still releasing another process's memory: try again
        if bodies.placement == "mixed" or args.placement == "plain":
            break
        bodies.free()
        m.lib().b3w_bodies_trim()                           # hand the pooled pieces back: the next search starts afresh
        time.sleep(1.0 + attempt)
        os.environ["B3W_PLACE_DEBUG"] = "1"                 # say on stderr what the search found
        bodies = ctx.alloc_bodies(n * pitch)
    d_bodies = bodies                                      # .data_ptr() like a tensor
    if args.variant is None:
        chosen, best_ms = ctx.autotune_device(d_recs.data_ptr(), n, bodies.ptr, pitch, d_pub.data_ptr(), d_status.data_ptr(),
                                              stream.cuda_stream)
    else:                                                   # (autotune leaves the winner selected in ctx)
        chosen = args.variant
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # HIP events on the launch stream around the K launches: average launch duration = region / K (event records
    # between the launches would cost 7 us per step: tools/ubench/launch_gap.py)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for i in range(args.steps):
        pub = ex.next_buffer()
        ctx.run_device(d_recs.data_ptr(), n, d_bodies.data_ptr(), pitch, pub.data_ptr(), d_status.data_ptr(),
                       stream.cuda_stream)
        ex.post()
    ev1.record(stream)
    allpub = ex.finish()                                    # every step's exchange is inside the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kern_ms = ev0.elapsed_time(ev1) / args.steps                      # HIP events on the launch stream
    assert int(d_status.abs().sum().item()) == 0, "a witness reported a non-zero status"
    # untimed: every body of the last step is checked on the device (recompute-from-own-inputs)
    d_mm = torch.full((n,), -1, dtype=torch.int32, device=dev)
    ctx.verify_device(d_bodies.data_ptr(), n, pitch, d_mm.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize()
    assert int(d_mm.abs().sum().item()) == 0, "on-device verification found a body that is not a valid witness"

    t = torch.tensor([elapsed, kern_ms], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed, kern_ms = t[0].item(), t[1].item()

    if rank == 0:
        total = world * n * args.steps
        alg_bytes = BYTES_PER_WITNESS[circuit] * n                   # per launch
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        traffic = None
        tf = os.path.join(ROOT, "profiles", "traffic_latest.json")   # PMC passes (WRITE_SIZE/FETCH_SIZE), see profiles/README.md
        if os.path.exists(tf):
            try:
                path = "sweep" if chosen >= 100 else "fused"
                for ent in json.load(open(tf)).get("entries", []):
                    if ent.get("circuit") == circuit and ent.get("batch") == n and ent.get("path") == path:
                        traffic = ent.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "BLAKE3-compression witnesses/sec",
            "value": total / elapsed,
            "unit": "witnesses/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": f"config2: batch {n} independent {circuit} witnesses per GPU, "
                                   f"{'BN254' if 'vesta' not in circuit else 'Vesta'} field, LCG(6429+i) inputs, "
                                   "device-resident inputs and outputs",
                       "circuit": circuit, "batch_per_gpu": n, "witness_bytes": ctx.body_bytes, "pitch": pitch,
                       "kernel_variant": "sweep (TRACE + SWEEP kernels)" if chosen >= 100 else f"fused ({chosen})",
                       "verified_on_device": True,
                       "placement": bodies.placement,
                       "exchange": "all_gather of public outputs (RCCL), pipelined with the next step's kernel" if world > 1 else "none"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": alg_bytes},
        }
        if args.cpu_seconds > 0 and world == 1:
            out["cpu_baseline"] = cpu_baseline(circuit, recs, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
