#!/usr/bin/env python3
"""bench.py — BLAKE3-compression witnesses/s on MI355X (BASELINE.json metric; N=1 workload = config 2).

A launch = one pass of the hot path over one batch: 4096 independent blake3_compression witnesses (BN254) per GPU,
inputs already resident in HBM, witness bodies written to HBM (3.16 GB per launch and GPU).  A launch lasts 0.44 ms,
so a timed STEP is `--inner` back-to-back launches over the same batch (default: as many as make the K timed steps
last >= 1 s; `config.launches_per_step` says how many) — sustained clocks, not a 9 ms burst.  With N > 1 GPUs every
rank runs its own 4096 instances (weak scaling; instance ids rank*4096 ...) and the ranks all-gather the per-witness
public outputs over RCCL after every launch, pipelined with the next launch.

  python bench.py [--gpus N] [--steps K] [--warmup W]          N > 1: spawns one fresh child process per GPU itself
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...      (the launcher form works too)

Prints ONE JSON line on rank 0 (see the driver contract) with `roofline` (HBM-write bound kernel: algorithmic
bytes / HIP-event kernel time) and, at N = 1, `cpu_baseline` (the C oracle timed on all host cores on a bounded
sample of the same workload, with the reference WASM's all-core rate beside it: live when --reference-dir holds the
reference, else as recorded in the build container by tools/wasm_baseline.py).
"""
import argparse
import contextlib
import importlib, json, math, os, sys, time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (about 6.3 TB/s achievable)
BYTES_PER_WITNESS = {"compression": 770976 + 112, "nova_bn254": 745312 + 128, "nova_vesta": 745312 + 128,
                     "nova_bn254_o1": 787648 + 128}   # body written + record read
FIELD = {"compression": "BN254", "nova_bn254": "BN254", "nova_vesta": "Vesta", "nova_bn254_o1": "BN254"}


def reference_wasm(circuit, reference_dir, seconds):
    """The reference's own WASM witness generator beside the GPU number: run live through tools/wasm_baseline.py
    — one node process per core looping the reference's calculateWTNSBin — when `reference_dir` holds the reference (it cannot
    travel to the GPU box, so that is the build container); otherwise the rates recorded there by the same tool
    (profiles/wasm_baseline.json), and the line says which."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import wasm_baseline as WB
        if seconds > 0 and circuit in WB.WASM and WB.available(reference_dir):
            r = WB.measure(circuit, reference_dir, seconds)
            return {"value": r["value"], "unit": "witnesses/s", "cores": r["cores"], "per_core": r["per_core"], "cpu": r["cpu"], "node": r["node"],
                    "measured_here": True, "sample": r["sample"], "where": f"this host, reference at {reference_dir}"}
    except Exception as e:                                     # a baseline leg never fails the bench
        print(f"bench.py: live WASM baseline failed ({e}); quoting the recorded one", file=sys.stderr)
    try:
        doc = json.load(open(os.path.join(ROOT, "profiles", "wasm_baseline.json")))
        r = doc["circuits"][circuit]
        return {"value": r["value"], "unit": "witnesses/s", "cores": r["cores"], "per_core": r["per_core"], "cpu": r["cpu"], "node": r["node"],
                "measured_here": False, "sample": r["sample"],
                "where": f"{doc['where']}, {doc['measured']}, profiles/wasm_baseline.json ({doc['tool']}); no reference checkout at {reference_dir} on this host"}
    except Exception:
        return None


WITNESS_KERNELS = ("b3w_compression_kernel", "b3w_nova_kernel", "b3w_sweep_kernel", "b3w_regionfill_kernel")


def kernel_path(v):
    """the kernel PATH a variant number stands for: the traffic per launch is a property of the path, not of the body-stream shape"""
    return "fused" if v is None or v < 100 else "sweep" if v < 200 else "fill"


def variant_name(v):
    return {"fused": f"fused ({v})", "sweep": "sweep (TRACE + SWEEP kernels)", "fill": f"fill-ordered fused (REGIONFILL, {v})"}[kernel_path(v)]
# store-only shapes of b3w_bodies_store_rate the batch line reads `achieved` against (roofline.store_ceiling)
STORE_SHAPES = {"streams_w4": 0, "streams_w8": 1, "fill": 2, "paced_persistent_w4x512": 3, "paced_persistent_w8x512": 4, "paced_streams_w8": 5,
                "paced_region_fill_sleep": 6, "paced_region_fill_valu": 7}
STORE_SHAPES_WHAT = ("b3w_bodies_store_rate: 20 passes of store-only kernels over the same n bodies — body streams (one wave per 4 / 8 bodies, 1 KiB per body "
                     "and step: the fused kernels' store shape), the runtime's fill shape, and the PACED shapes (four vector-ALU instructions in front of every "
                     "store; 512 persistent waves taking groups of 4 / 8 bodies, or one wave per 8 bodies) that the round-6 sweep of 100 shapes found fastest "
                     "(tools/ubench/store_sweep.hip, profiles/r06/store_sweep*.log), and the fill-ordered kernel's store order paced by s_sleep / by vector-ALU "
                     "instructions (tools/ubench/store_region_scan.py): an unpaced store-only kernel fills HBM slower than the witness kernel")


def kernel_row_path(name):
    """the kernel PATH a dispatch of the counter passes belongs to (None: not a timed witness kernel): the body-stream kernels' MODE-0
    instantiations are `fused`; the fill-ordered kernel and — nova — the wide-slot launch behind it (MODE 3) are `fill`"""
    import re
    if "b3w_regionfill_kernel" in name:
        return "fill"
    m = re.search(r"(?:true|false), (\d), (?:true|false)(?:, (?:true|false))?>\(", name)
    if m and ("b3w_compression_kernel" in name or "b3w_nova_kernel" in name):
        return {"0": "fused", "3": "fill"}.get(m.group(1))
    return None


def live_traffic(args):
    """HBM bytes per launch of the witness kernel(s), measured for THIS invocation: two child runs of this script under
    `rocprofv3 --pmc WRITE_SIZE` and `--pmc FETCH_SIZE` (separate passes, as MI355X_MICROARCH.md's HBM section prescribes; values are
    KiB; on gfx950 FETCH_SIZE reports half of a wide read stream: doubled), BEFORE this process touches the GPU.  A child (`--traffic-child`)
    launches the batch 8 times through EACH kernel path the parent's autotune may end on — the body streams (variant 0: all fused variants
    move the same bytes) and, where the circuit has it, the fill-ordered kernel —; the parent takes the path it timed.
    -> ({path: {...}}, None) or (None, why not)."""
    import csv, glob, shutil, subprocess, tempfile
    roc = shutil.which("rocprofv3")
    if not roc:
        return None, "no rocprofv3 on PATH"
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "this run is itself under a profiler"
    tmp = tempfile.mkdtemp(prefix="b3w_traffic_", dir="/tmp")
    per, t0 = {}, time.perf_counter()
    try:
        for counter in ("WRITE_SIZE", "FETCH_SIZE"):
            # (a plain buffer: the bytes a kernel moves do not depend on where its buffer lies, and the child needs no placement search)
            cmd = [roc, "--pmc", counter, "--output-format", "csv", "-d", os.path.join(tmp, counter), "--", sys.executable, os.path.abspath(__file__),
                   "--traffic-child", "1", "--circuit", args.circuit, "--batch", str(args.batch), "--pitch", str(args.pitch), "--placement", "plain",
                   "--cpu-seconds", "0"] + (["--variant", str(args.variant)] if args.variant is not None else [])
            r = subprocess.run(cmd, cwd=tmp, env=dict(os.environ, TMPDIR="/tmp", B3W_PLACE_CHECK="0"), capture_output=True, text=True, timeout=120)     # (2 s a pass when warm; the first child of a fresh box also pays the cold `import torch`)
            if r.returncode != 0:
                return None, f"the {counter} pass failed (rc {r.returncode}): " + (r.stderr or r.stdout)[-200:].replace("\n", " | ")
            child = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
            files = glob.glob(os.path.join(tmp, counter, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                return None, f"the {counter} pass left no counter_collection.csv"
            rows = [x for x in csv.DictReader(open(files[0])) if x["Counter_Name"] == counter]
            rows.sort(key=lambda x: int(x["Dispatch_Id"]))
            for path, kernels_per_launch in child["paths"].items():
                mine = [x for x in rows if kernel_row_path(x["Kernel_Name"]) == path]
                k = child["launches"] * kernels_per_launch            # the child's last launches of that path
                if len(mine) < k:
                    return None, f"the {counter} pass shows {len(mine)} dispatches of the {path} path, {k} expected"
                per.setdefault(path, {})[counter] = sum(float(x["Counter_Value"]) for x in mine[-k:]) * 1024.0 / child["launches"]
    except Exception as e:                                  # (a time-out, an unreadable csv: the line then quotes the recorded passes)
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    secs = round(time.perf_counter() - t0, 1)
    return {path: {"hbm_bytes_per_launch": c["WRITE_SIZE"] + 2.0 * c["FETCH_SIZE"], "write_bytes": c["WRITE_SIZE"], "fetch_bytes_x2": 2.0 * c["FETCH_SIZE"],
                   "seconds": secs, "path": path} for path, c in per.items() if len(c) == 2}, None


def self_launch(n, argv, launch_timeout):
    """`python bench.py --gpus N` without a launcher: N fresh child processes, one rank per GPU, started BEFORE this
    process has touched HIP (it never does: the parent only waits).  Rank 0's child prints the JSON line; the parent
    exits with the first non-zero child status, after ending the other children (they would wait in a collective).
    Watchdog: every rank reports "torch imported" and "rendezvous passed" through files in a scratch directory; when a rank
    has not passed rendezvous `launch_timeout` seconds after the first one finished importing torch (a rank stuck in
    ncclCommInitRank, a peer that never came up), all children are ended and the exit status is 124.  The children run in
    their own sessions: the parent forwards SIGTERM / SIGINT to them and leaves no rank behind, whatever ends it."""
    import shutil, signal, socket, subprocess, tempfile
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    stage = tempfile.mkdtemp(prefix="b3w_bench_")
    procs = []

    def end_all(sig):
        for q in procs:
            if q.poll() is None:
                try:
                    os.killpg(q.pid, sig)                   # exactly the sessions started below (start_new_session)
                except (ProcessLookupError, PermissionError):
                    pass

    def on_signal(signum, frame):
        end_all(signal.SIGTERM)
        raise SystemExit(128 + signum)
    old = {sg: signal.signal(sg, on_signal) for sg in (signal.SIGTERM, signal.SIGINT)}
    rc = 0
    try:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), B3W_BENCH_STAGE_DIR=stage)
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # this pool's host driver only does dmabuf IPC
            env.setdefault("OMP_NUM_THREADS", "1")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env, start_new_session=True))
        live = list(procs)
        t_start, t_first_import, t_fail = time.monotonic(), None, None
        while live:
            time.sleep(0.1)
            if rc != 0:
                t_fail = t_fail or time.monotonic()
                if time.monotonic() - t_fail > 5.0:          # ranks that do not react to SIGTERM: the `finally` below kills them
                    break
            for p in list(live):
                st = p.poll()
                if st is None:
                    continue
                live.remove(p)
                if st != 0 and rc == 0:
                    rc = st if st > 0 else 128 - st
                    end_all(signal.SIGTERM)
            if rc == 0 and live and launch_timeout > 0:
                names = os.listdir(stage)
                if t_first_import is None and any(x.startswith("imported.") for x in names):
                    t_first_import = time.monotonic()
                ready = sum(1 for x in names if x.startswith("ready."))
                now = time.monotonic()
                late = (t_first_import is not None and now - t_first_import > launch_timeout) or now - t_start > launch_timeout + 300
                if ready < n and late:
                    missing = sorted(set(range(n)) - {int(x.split(".")[1]) for x in names if x.startswith("ready.")})
                    print(f"bench.py: ranks {missing} have not passed rendezvous after {launch_timeout:g} s "
                          f"(--launch-timeout); ending all {n} ranks", file=sys.stderr, flush=True)
                    rc = 124
                    end_all(signal.SIGTERM)
        return rc
    finally:
        t_kill = time.monotonic() + 5.0                      # a rank that ignores SIGTERM (stuck in a driver call) is killed
        while any(q.poll() is None for q in procs) and time.monotonic() < t_kill:
            end_all(signal.SIGTERM)
            time.sleep(0.2)
        end_all(signal.SIGKILL)
        for q in procs:
            try:
                q.wait(timeout=5)
            except Exception:
                pass
        for sg, h in old.items():
            signal.signal(sg, h)
        shutil.rmtree(stage, ignore_errors=True)


def stage_mark(name, rank):
    """tell the launching parent (self_launch) how far this rank has come"""
    d = os.environ.get("B3W_BENCH_STAGE_DIR")
    if d:
        try:
            open(os.path.join(d, f"{name}.{rank}"), "w").close()
        except OSError:
            pass


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(circuit, recs, budget_s, reference_dir="/root/reference", reference_s=10.0):
    """Time the oracle (oracle/libb3w_oracle.so, the CPU restatement = "port") on ALL host cores this process may
    use, over a bounded sample of the same records.  Checker code, used here only as the reported baseline; the
    reference WASM cannot travel to the GPU box, so its build-container rate is quoted."""
    import ctypes, threading
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import b3w_testlib as T
    import numpy as np
    lib = T.oracle()                                          # ctypes releases the GIL inside the C call
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:                                                      # a container's CPU share (cgroup v2): "max" or "<quota> <period>"
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(per)
            cores = max(1, min(cores, math.ceil(quota)))
    except (OSError, ValueError):
        pass
    sample = np.ascontiguousarray(recs[:16], dtype=np.uint32)  # 12 MB of bodies per thread, rewritten in place (cache resident)
    k, cid = sample.shape[0], T.CIRCUIT_ID[circuit]
    bufs = [np.zeros((k, T.NWIT[circuit] * 32), dtype=np.uint8) for _ in range(cores)]
    for b in bufs[:1]:
        assert lib.b3wo_witness_batch_u32(cid, sample.ctypes.data, k, b.ctypes.data) == 0      # warm-up
    done = [0] * cores
    t_end = [0.0] * cores
    t0 = time.perf_counter()

    def work(i):
        while time.perf_counter() - t0 < budget_s:
            assert lib.b3wo_witness_batch_u32(cid, sample.ctypes.data, k, bufs[i].ctypes.data) == 0
            done[i] += k
        t_end[i] = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = max(t_end) - t0
    total = sum(done)
    out = {"value": total / dt, "unit": "witnesses/s", "cores": cores, "kind": "port", "cpu": cpu_model(),
           "per_core": total / dt / cores, "cpu_quota": quota,
           "sample": f"{total} witnesses in {dt:.1f} s: {cores} threads, each a cache-resident {k}-record loop (the first {k} records "
                     f"of the workload, rewriting one 12 MB block of bodies in place), C oracle (oracle/b3w_oracle.c)"}
    ref = reference_wasm(circuit, reference_dir, reference_s)
    if ref is not None:
        out["reference_wasm"] = ref
    return out


def native_comm(m, ctx, dist, world, rank, tag):
    """--exchange-impl native: a b3w_comm over the ranks of the torch process group — RCCL (the id travels through the group)
    under "nccl", the host shared-memory transport under "gloo" (several ranks on one GPU)."""
    if dist.get_backend() == "nccl":
        box = [m.Comm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return m.Comm(ctx, box[0], rank, world)
    return m.Comm.host(ctx, f"/b3w_bench_{os.environ.get('MASTER_PORT', '0')}_{tag}", rank, world)


def valu_ceiling():
    """Field multiplications per second of the chip in the commit kernel's arithmetic (nine 29-bit limbs), multiplications only:
    tools/ubench/fpmul29_peak.hip, latest run under profiles/ (G/s for mul and sqr).  A mixed addition is 8 mul + 2 sqr."""
    mul, sqr, src = 172.8, 199.1, "profiles/r01/commit/fpmul29_peak.log"
    try:
        doc = json.load(open(os.path.join(ROOT, "profiles", "valu_ceiling_latest.json")))
        mul, sqr, src = float(doc["mul_g_per_s"]), float(doc["sqr_g_per_s"]), doc.get("source", src)
    except Exception:
        pass
    return 10.0 / (8.0 / mul + 2.0 / sqr), src


@contextlib.contextmanager
def native_stdout_to_stderr():
    """stdout is ONE JSON line: what a native library writes to fd 1 meanwhile (gloo's "[Gloo] Rank 0 is connected to ..." at
    rendezvous) goes to stderr."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


def placement_cost(ctx, before=None):
    c = ctx.placement_cost()
    if before:
        for k in ("search_s", "search_gib_walked", "check_s", "search_timeouts"):
            c[k] -= before[k]
    return {"placement_search_s": round(c["search_s"], 3), "placement_search_gib_walked": round(c["search_gib_walked"], 2),
            "placement_check_s": round(c["check_s"], 3), "placement_search_timeouts": int(c["search_timeouts"]),
            "placement_search_limit_s": c["search_limit_s"],
            "placement_search_breakdown_s": {"hipMemCreate": round(c["create_s"], 3), "map": round(c["map_s"], 3), "probes": round(c["probe_s"], 3),
                                             "release": round(c["release_s"], 3)}}


def gather_strings(dist, world, s):
    if world == 1:
        return [s]
    out = [None] * world
    dist.all_gather_object(out, s)
    return out


def bench_chain(args, m, torch, dist, dev, world, rank, local_rank, devices=None):
    """Configs 4/5: a step = one pass over the whole preimage (plan + all leaf and parent step witnesses of this
    rank's chunk range, bodies streamed through a ring of batch buffers).  Strong scaling: the preimage is fixed.
    Preimage = little-endian byte stream of LCG(1)."""
    circuit = args.circuit if args.circuit != "compression" else "nova_vesta"
    ctx = m.Context(circuit, local_rank)
    nbytes = int(args.preimage_mib * (1 << 20))
    host = torch.from_numpy(m.workloads.lcg_preimage(nbytes, seed=1).copy()).pin_memory()
    consumer, key, commit_only, commit_records, r1cs_t = None, None, None, None, None
    n_max = m.lib().b3w_chain_num_chunks(nbytes) * 64 + 64
    if "check" in args.consumer:                     # SURVEY.md 8(f) row 2, first half: Az * Bz = Cz for every step witness, timed
        if circuit not in m.BUILTIN_R1CS:
            raise SystemExit(f"bench.py: no constraint system for {circuit}")
        r1cs_t = m.R1cs(ctx)                         # the chain checks every batch itself (b3w_chain_check_constraints): fold_witnesses(check=)
    if "commit" in args.consumer:                    # second half: what the folding prover does with each step witness
        K = importlib.import_module("hot-proofs-blake3-circom_amd.synthetic_key")
        curve = "vesta" if "vesta" in circuit else "bn254_g1"
        # (folded key: slots the circuit's linear constraints express through others drop out of the tables — the circomkit /
        # compression builds lose half their virtual slots, the O2 builds have no linear constraints left: DESIGN.md 8d)
        key = m.CommitKey(ctx, curve, K.generators(curve, ctx.witness_size, seed=b"bench"), window=args.commit_window, fold=True if circuit in m.BUILTIN_R1CS else None)
        d_pts = torch.zeros((n_max, 64), dtype=torch.uint8, device=dev)
        d_st = torch.zeros(n_max, dtype=torch.int32, device=dev)
        if args.consumer == "commit-only":
            commit_only = (key, d_pts)
        elif "bodies" not in args.consumer:          # commit / check+commit: the commitments come from the step records, beside the bodies
            commit_records = (key, d_pts)
        else:                                        # commit-bodies: the kernel that READS the bodies (foreign bodies), behind the check
            def consumer(view, first, k):
                key.commit_device(view.data_ptr(), k, view.stride(0), d_pts.data_ptr() + 64 * first, d_st.data_ptr() + 4 * first,
                                  torch.cuda.current_stream().cuda_stream)
    # the fold's exchange (N > 1) is part of every pass: chunk chaining values, then every step's h_out (BASELINE config 4)
    comm = native_comm(m, ctx, dist, world, rank, "chain") if (world > 1 and args.exchange_impl == "native") else None
    # steps per ring buffer (`--batch`, 4096 = not given): 32 768 where a constraint check is in the pass — its per-batch costs (planner, TRACE,
    # normalisation, the launches' gaps) are paid half as often: +2.5 % `check`, +3.6 % `check+commit` — else 16 384: the commitments that run FREE
    # beside the witness kernels pipeline better over more, smaller batches (4.76 against 4.57 M steps/s); the ring is 2 x 24 or 2 x 12 GB
    batch_steps = args.batch if args.batch != 4096 else 32768 if "check" in args.consumer else 16384
    run = lambda: m.chain.fold_witnesses(ctx, host, batch_steps=batch_steps, ring=2, consumer=consumer,
                                         commit_only=commit_only, commit_records=commit_records, gather_hout=args.exchange != "none", comm=comm,
                                         check=r1cs_t, commit_overlap=args.commit_overlap)
    t_first = time.perf_counter()
    for i in range(max(3, args.warmup)):        # the first passes pay the allocator (24 GB ring, record buffers)
        out = run()
        if i == 0:
            torch.cuda.synchronize()
            first_pass_s = time.perf_counter() - t_first
    torch.cuda.synchronize()
    place_cost = placement_cost(ctx)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = run()
    torch.cuda.synchronize()
    local_elapsed = time.perf_counter() - t0                # this rank's own passes, before it waits for the others
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    assert int(out["status"].abs().sum().item()) == 0
    ex_ms = m.chain.exchange_ms(out)                          # the last timed pass: (chunk CVs, h_out) on this rank's device
    recs16 = out["records"][:16].cpu().numpy().copy()         # (the views die with the chain object: the verification pass below may replace it)
    local_steps = out["n_leaf_steps"] + out["n_parent_steps"]
    n_leaf_all = m.lib().b3w_chain_num_leaf_steps(nbytes)
    if args.exchange != "none" or world == 1:
        # the gathered h_out, on every rank: the last leaf step of chunk c carries chunk c's chaining value — checked for this
        # rank's own chunks against the planner's CVs, and the array has every rank's rows
        h_all = out["h_out_all"]
        assert h_all.shape == (n_leaf_all, 8)
        c0, ncl = out["first_chunk"], out["n_chunks_local"]
        full = ncl - (1 if (c0 + ncl == out["n_chunks"] and nbytes % 1024) else 0)
        if full > 0:
            assert torch.equal(h_all[c0 * 16 + 15:(c0 + full) * 16:16], out["chunk_cvs_local"][:full]), "gathered h_out of step 16c+15 is not chunk c's chaining value"
        if world > 1:
            assert int((h_all.abs().sum(dim=1) == 0).sum().item()) == 0, "rows of another rank are missing from the gathered h_out"
            # every rank has checked its OWN rows against its planner; the same array (and the same root, built from the gathered chunk
            # CVs) on every rank then means every rank holds every other rank's rows as that rank computed them
            w64 = h_all.to(torch.int64) & 0xFFFFFFFF
            pos = (torch.arange(h_all.shape[0], device=dev, dtype=torch.int64) % 65521 + 1).unsqueeze(1)
            sig = f"{int(w64.sum().item())}:{int((w64 * pos).sum().item())}:{out['root'].cpu().tolist()}"
            sigs = gather_strings(dist, world, sig)
            assert len(set(sigs)) == 1, f"the ranks disagree about the gathered h_out or the root: {sigs}"
    # untimed: one more pass over the first MiB (at most) of the preimage whose consumer checks EVERY step witness against the
    # step circuit's rank-1 constraints while it sits in the ring
    verification = "none"
    if circuit in m.BUILTIN_R1CS and commit_only is None and r1cs_t is None:
        r1cs = m.R1cs(ctx)
        vsteps = [0]
        d_viol = torch.zeros(16384, dtype=torch.int32, device=dev)
        d_sum = torch.zeros(1, dtype=torch.int64, device=dev)

        def check(view, first, k):
            r1cs.check_device(view.data_ptr(), k, view.stride(0), d_viol.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
            d_sum.add_(d_viol[:k].to(torch.int64).sum())
            vsteps[0] += k
        m.chain.fold_witnesses(ctx, host[:min(nbytes, 1 << 20)].clone().pin_memory() if nbytes > (1 << 20) else host, batch_steps=16384, ring=2,
                               consumer=check)
        torch.cuda.synchronize()
        assert int(d_sum.item()) == 0, "a step witness violates the step circuit's rank-1 constraints"
        verification = f"r1cs: 0 of {r1cs.n_constraints} constraints violated by any of {vsteps[0]} step witnesses of the first MiB (rank 0's share)"
        r1cs.close()
    if key is not None:
        assert int(d_st[:local_steps].abs().sum().item()) == 0 and int(d_pts[:local_steps].max(dim=1).values.min().item()) > 0
        if commit_records is not None and r1cs_t is not None:
            # the commitments of the fold-shaped pass against the kernel that READS bodies: the first 256 leaf steps' bodies, regenerated
            kk = min(256, local_steps)
            d_b = torch.empty((kk, ctx.body_bytes), dtype=torch.uint8, device=dev)
            d_p2 = torch.zeros((kk, 64), dtype=torch.uint8, device=dev)
            ctx.run_device(out["records"].data_ptr(), kk, d_b.data_ptr(), 0, 0, 0, torch.cuda.current_stream().cuda_stream)
            key.commit_device(d_b.data_ptr(), kk, ctx.body_bytes, d_p2.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            assert torch.equal(d_p2, d_pts[:kk]), "commitments from the records differ from commitments of the bodies"
    if r1cs_t is not None:
        assert int(out["violations"][:local_steps].abs().sum().item()) == 0, "a step witness violates the step circuit's rank-1 constraints"
        verification = f"r1cs, inside the timed pass: 0 of {r1cs_t.n_constraints} constraints violated by any of {local_steps} step witnesses (rank 0's share)"
    t = torch.tensor([elapsed, float(local_steps)], dtype=torch.float64, device=dev)
    if world > 1:
        tm = t.clone(); dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        ts = t.clone(); dist.all_reduce(ts, op=dist.ReduceOp.SUM)
        elapsed, total_steps = tm[0].item(), ts[1].item()
    else:
        total_steps = float(local_steps)
    placements = gather_strings(dist, world, out.get("placement"))
    pass_ms = [float(x) for x in gather_strings(dist, world, repr(local_elapsed / args.steps * 1e3))]
    ex_all = [json.loads(x) for x in gather_strings(dist, world, json.dumps([round(ex_ms[0], 4), round(ex_ms[1], 4)]))]
    # The roofline of the pass is its dominant kernel's.  No consumer: the witness kernel, HBM writes.  check: witness kernel +
    # constraint check, both HBM-bound — every body is written once and read once.  commit / commit-only: the commit kernel, bound by
    # the vector ALUs: mixed point additions of 8 field multiplications + 2 squarings, priced against the multiplication-only ceiling
    # of tools/ubench/fpmul29_peak.hip; the additions are COUNTED by the kernel in one extra, untimed pass.
    body_b = 32 * ctx.witness_size
    rate = total_steps * args.steps / elapsed / world                         # steps per second and GPU
    if key is None:
        per = BYTES_PER_WITNESS[circuit] + (body_b if r1cs_t is not None else 0)
        roof = {"bound": "hbm", "achieved": rate * per / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": rate * per / 1e9 / HBM_PEAK_GBS, "traffic": None,
                "algorithmic_bytes_per_step": per,
                "note": ("bytes per step = body written + record read" + (" + body read back by the constraint check" if r1cs_t is not None else "")
                         + "; end-to-end per-GPU rate incl. planner, H2D and launch gaps")}
    else:
        key.count(True)
        run()
        torch.cuda.synchronize()
        adds, wits = key.counts()
        key.count(False)
        adds_per_step = adds / max(wits, 1)
        tree = 63 * 14 + 400                                                  # LDS tree (63 full additions of 12 M + 2 S) + normalisation (one Fermat inversion)
        mults = 10.0 * adds_per_step + tree
        peak, src = valu_ceiling()
        roof = {"bound": "valu", "achieved": rate * mults / 1e9, "peak": peak, "unit": "G field mul/s", "frac": rate * mults / 1e9 / peak, "traffic": None,
                "field_multiplications_per_step": mults, "point_additions_per_step": adds_per_step, "peak_source": src,
                "note": "29-bit-limb field multiplications of the commit kernel (10 per mixed addition, counted by the kernel in an untimed pass; "
                        "+ tree and normalisation) against the chip's multiplication-only rate; end-to-end per-GPU rate"
                        + ("" if commit_only is not None else ", witness generation" + (" and constraint check" if r1cs_t is not None else "") + " included in the time"
                           + ("" if commit_records is not None else "; this consumer also reads every body back"))}
    cpu = None
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        import numpy as np
        cpu = cpu_baseline(circuit, recs16.view(np.uint32), args.cpu_seconds, args.reference_dir, args.reference_seconds)
    if rank == 0:
        line = {
            "metric": "BLAKE3-compression witnesses/sec", "value": total_steps * args.steps / elapsed, "unit": "witnesses/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"config{5 if args.preimage_mib >= 1024 else 4}-style chain: {args.preimage_mib:g} MiB preimage "
                                   f"(LE stream of LCG(1)) -> {int(total_steps)} nova steps ({circuit}), "
                                   "planner + witness kernels, bodies through a 2-deep ring, H2D overlapped",
                       "circuit": circuit, "n_chunks": out["n_chunks"], "path_len": out["path_len"], "batch_steps": batch_steps, "ring": 2,
                       "placement": placements[0], "placement_per_rank": placements, "verification": verification,
                       "exchange": "none" if world == 1 else
                                   (f"all_gather of {n_leaf_all} x 8 u32 h_out (+ {int(total_steps) - n_leaf_all} x 8 of the parent steps) + "
                                    if args.exchange != "none" else "--exchange none: ") +
                                   f"all_gather of {out['n_chunks']} x 8 u32 chunk chaining values over {dist.get_world_size()} ranks "
                                   f"({dist.get_backend()}), inside every timed pass",
                       "exchange_mode": ("every" if args.exchange != "none" else "none") if world > 1 else "none",
                       "exchange_impl": ("native b3w_comm (" + comm.transport + ")" if comm is not None else "torch.distributed") if world > 1 else "none",
                       "comm_size": (int(m.lib().b3w_comm_size(comm.handle)) if comm is not None else dist.get_world_size()) if world > 1 else 1,
                       "exchange_ms_per_rank": {"chunk_cvs": [x[0] for x in ex_all], "h_out": [x[1] for x in ex_all],
                                                "what": "HIP events on the compute stream around staging + collective + scatter, last timed pass"},
                       "pass_ms_per_rank": pass_ms, "first_pass_s": round(first_pass_s, 3), "devices_per_rank": devices or [], **place_cost,
                       "commit_overlap": args.commit_overlap if commit_records is not None else None,
                       "consumer": " then ".join(
                           ([f"rank-1 constraint check of every step witness on the device ({r1cs_t.n_constraints} constraints)"] if r1cs_t is not None else []) +
                           ([f"Pedersen commitment of every step witness on the device ({key.window}-bit windows, {key.folded_slots} slots folded)"
                             + (", from the step records: no bodies written" if commit_only is not None else
                                ", from the step records, beside the bodies" if commit_records is not None else ", read from the bodies")] if key is not None else [])) or "none"},
            "roofline": roof,
        }
        if cpu is not None:
            line["cpu_baseline"] = cpu
        print(json.dumps(line), flush=True)
    if comm is not None:
        comm.close()
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--inner", type=int, default=0,
                    help="batch workload: launches per timed step (0 = as many as make the timed region last >= 5 s)")
    ap.add_argument("--batch", type=int, default=4096, help="witnesses per GPU per launch")
    ap.add_argument("--circuit", default="compression")
    ap.add_argument("--variant", type=int, default=None, help="kernel tuning variant (B3W_VARIANT)")
    ap.add_argument("--pitch", type=int, default=0, help="body pitch in bytes (0 = contiguous bodies)")
    ap.add_argument("--cpu-seconds", type=float, default=5.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--reference-dir", default="/root/reference",
                    help="cpu_baseline: where the reference checkout lies.  When it is there (the build container) its WASM witness generator "
                         "is timed live on all host cores (tools/wasm_baseline.py); when not (the GPU box: the reference cannot travel) the "
                         "rates recorded by the same tool in profiles/wasm_baseline.json are quoted, and the line says so")
    ap.add_argument("--reference-seconds", type=float, default=10.0, help="budget of that live leg (0 = always quote the recorded rates)")
    ap.add_argument("--placement", default="mixed", choices=["mixed", "plain"],
                    help="body buffer placement: mixed = b3w_bodies_alloc's two-class buffer (default), plain = hipMalloc")
    ap.add_argument("--workload", default="batch", choices=["batch", "chain"],
                    help="batch = BASELINE config 2/3 (default, the headline metric); chain = configs 4/5: "
                         "preimage -> planner -> nova step witnesses, streamed through a ring of buffers")
    ap.add_argument("--preimage-mib", type=float, default=1.0, help="chain workload: preimage size (1 = config 4, 1024 = config 5)")
    ap.add_argument("--consumer", default="none", choices=["none", "commit", "commit-only", "check", "check+commit", "commit-bodies", "check+commit-bodies"],
                    help="chain workload: what is done with each batch of step witnesses while it sits in the ring.  check = the step "
                         "circuit's rank-1 constraints over every step witness; commit = Pedersen commitments on the circuit's curve "
                         "(synthetic generators) computed from the step records beside the bodies (b3w_chain_commit_from_records: the "
                         "fold-shaped pass); commit-only = the same commitments, no bodies written; check+commit = both; "
                         "commit-bodies / check+commit-bodies = the commitment kernel that READS the bodies (for bodies the library did not make)")
    ap.add_argument("--commit-window", type=int, default=0, choices=[0, 12, 16, 18],
                    help="chain workload with a commit consumer: bits per table window of the commitment key (0 = the library's choice)")
    ap.add_argument("--commit-overlap", default="auto", choices=["auto", "serial", "free", "gated"],
                    help="chain workload, commit / check+commit: where the commitments from the records run (b3w_chain_commit_overlap): on the "
                         "caller's stream, free on the chain's commit stream, or gated to the batch's own witness kernel; auto = gated when a "
                         "check or consumer reads the batch, else free")
    ap.add_argument("--exchange", default="every", choices=["every", "last", "none"],
                    help="N > 1: the fold's exchange inside the timed region.  batch workload: all-gather of the public outputs "
                         "after EVERY launch (default; pipelined with the next launch), after the LAST launch only, or not at all — "
                         "the three together split a scaling number into kernel and RCCL contention.  chain workload: every / last "
                         "= gather every step's h_out inside each pass, none = chunk chaining values only")
    ap.add_argument("--exchange-impl", default="torch", choices=["torch", "native"],
                    help="N > 1: who runs the exchange.  torch = torch.distributed collectives (nccl = RCCL; gloo staged through the host); "
                         "native = the C-ABI's own b3w_comm (RCCL through librccl under nccl; the host shared-memory transport under "
                         "B3W_DIST_BACKEND=gloo, several ranks on one GPU): b3w_chain_run_parents_sharded + b3w_chain_allgather_hout in the "
                         "chain workload, b3w_comm_allgather of the public outputs in the batch workload")
    ap.add_argument("--placement-search-s", type=float, default=-1.0,
                    help="time limit of the placement allocator's search for a second class of HBM (then the buffer is plain and says so); "
                         "default: the library's own (b3w_bodies_search_limit)")
    ap.add_argument("--placement-search-gib", type=int, default=-1,
                    help="GiB of new memory a placement search may touch beyond the buffer; default: the library's own (b3w_bodies_configure)")
    ap.add_argument("--launch-timeout", type=float, default=120.0,
                    help="N > 1: seconds every rank has to pass rendezvous (process group + first barrier), counted from the moment "
                         "the first rank has imported torch; also the process group's own timeout.  0 = no watchdog")
    ap.add_argument("--timed-ms", type=float, default=5000.0,
                    help="batch workload with --inner 0: how long the K timed steps should last together (the driver's busy "
                         "sampler needs seconds, not a 9 ms burst)")
    ap.add_argument("--traffic", default="auto", choices=["auto", "quoted"],
                    help="roofline.traffic of the batch line at N = 1: auto = measured by this invocation (two rocprofv3 --pmc child passes of 8 "
                         "launches, before anything else; a few seconds) where rocprofv3 is there and this run is not itself profiled, "
                         "else — and with `quoted` — the figure of the recorded passes (profiles/traffic_latest.json), said so in traffic_source")
    ap.add_argument("--traffic-child", default=None, help=argparse.SUPPRESS)      # (internal: what --traffic auto runs under rocprofv3)
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus, sys.argv[1:], args.launch_timeout))      # nothing above has loaded torch or HIP
    measured_traffic, traffic_why_not = None, "--traffic quoted"
    if args.traffic == "auto" and args.traffic_child is None and args.workload == "batch" and "WORLD_SIZE" not in os.environ:
        measured_traffic, traffic_why_not = live_traffic(args)      # (children of a process that has not touched the GPU yet)

    rank = int(os.environ.get("RANK", "0"))
    hang = os.environ.get("B3W_BENCH_TEST_HANG_RANK")       # tests only: this rank never reaches the rendezvous
    if hang is not None and str(rank) in hang.split(",") and os.environ.get("B3W_BENCH_TEST_HANG_AT", "start") == "start":
        time.sleep(3600)

    import datetime
    import numpy as np
    import torch
    import torch.distributed as dist
    stage_mark("imported", rank)
    if hang is not None and str(rank) in hang.split(",") and os.environ.get("B3W_BENCH_TEST_HANG_AT") == "imported":
        time.sleep(3600)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # this pool's host driver only supports dmabuf IPC: with the legacy mode RCCL's (and torch's) cross-process buffer
        # sharing fails in hipIpcGetMemHandle.  Left alone when the environment already sets it.
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.gpus != world and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} ranks; using {world}", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product has no CPU path")
    ndev = torch.cuda.device_count()
    backend = os.environ.get("B3W_DIST_BACKEND", "nccl")      # "nccl" = RCCL over xGMI; "gloo" only for dry runs
    if local_rank >= ndev:
        if backend == "nccl":
            raise SystemExit(f"bench.py: rank {rank} has no GPU of its own ({ndev} visible, one rank per GPU); "
                             "B3W_DIST_BACKEND=gloo rehearses several ranks on one GPU")
        local_rank = local_rank % ndev                      # dry run: several ranks share one GPU
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    devices = [f"cuda:{local_rank}"]
    if world > 1:
        if hang is not None and str(rank) in hang.split(",") and os.environ.get("B3W_BENCH_TEST_HANG_AT") == "rendezvous":   # tests only: hung with the GPU initialised
            time.sleep(3600)
        # (the process group's own limit is the backstop under a foreign launcher; 30 s later than the watchdog so that it is the
        # watchdog that reports which ranks were missing)
        pg_timeout = datetime.timedelta(seconds=args.launch_timeout + 30 if args.launch_timeout > 0 else 1800)
        with native_stdout_to_stderr():
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=pg_timeout)
            else:
                dist.init_process_group(backend, rank=rank, world_size=world, timeout=pg_timeout)
        # one rank per GPU: every rank's card must be a different one (uuid where torch reports it, else the PCI address)
        pr = torch.cuda.get_device_properties(dev)
        ident = f"{getattr(pr, 'uuid', '')}@pci{getattr(pr, 'pci_domain_id', 0):04x}:{getattr(pr, 'pci_bus_id', 0):02x}:{getattr(pr, 'pci_device_id', 0):02x}"
        devices = gather_strings(dist, world, f"cuda:{local_rank}={ident}")
        if backend == "nccl" and len(set(d.split("=", 1)[1] for d in devices)) != world:
            raise SystemExit(f"bench.py: ranks share a GPU under RCCL: {devices}")
        dist.barrier()
    stage_mark("ready", rank)

    if args.variant is not None:
        os.environ["B3W_VARIANT"] = str(args.variant)
    if args.placement == "plain":                            # (both workloads: the library's allocator reads it)
        os.environ["B3W_PLACEMENT"] = "plain"
    m = importlib.import_module("hot-proofs-blake3-circom_amd")
    W = m.workloads
    # placement runs on the LIBRARY's defaults (what an integrator of the C-ABI gets); --placement-search-gib / -s are experiments
    if args.placement_search_gib >= 0:
        m.lib().b3w_bodies_configure(args.placement_search_gib, -1)
    if args.placement_search_s >= 0:
        m.lib().b3w_bodies_search_limit(args.placement_search_s)
    if args.workload == "chain":
        return bench_chain(args, m, torch, dist, dev, world, rank, local_rank, devices)
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
    # of launch i overlaps the kernel of launch i+1 on RCCL's own stream (sharding.PublicExchange)
    ex = sharding.PublicExchange(n, npub, dev)
    comm = None
    if world > 1 and args.exchange_impl == "native":
        comm = native_comm(m, ctx, dist, world, rank, "batch")

        class NativeExchange:
            """the same hand-over as sharding.PublicExchange, through b3w_comm_allgather on the launch stream"""
            def __init__(self):
                self.bufs = [torch.zeros((n, npub), dtype=torch.int32, device=dev) for _ in range(2)]
                self.outs = [torch.empty((world * n, npub), dtype=torch.int32, device=dev) for _ in range(2)]
                self.i, self.last = 0, None

            def next_buffer(self):
                return self.bufs[self.i % 2]

            def post(self):
                k = self.i % 2
                comm.allgather(self.bufs[k].data_ptr(), self.outs[k].data_ptr(), n * npub * 4, stream.cuda_stream)
                self.last = k
                self.i += 1

            def finish(self):
                return None if self.last is None else self.outs[self.last]
        ex = NativeExchange()

    def launch(post=None):
        pub = ex.next_buffer()
        ctx.run_device(d_recs.data_ptr(), n, d_bodies.data_ptr(), pitch, pub.data_ptr(), d_status.data_ptr(),
                       stream.cuda_stream)
        if post if post is not None else args.exchange == "every":
            ex.post()

    # Untimed set-up.  The body buffer comes from the library's placement allocator (b3w_bodies_alloc: its 256 MiB
    # pieces alternate between two classes of HBM, DESIGN.md "Placement"), then the faster of the bit-identical
    # kernel variants is picked on that buffer.
    if args.placement == "plain":
        os.environ["B3W_PLACEMENT"] = "plain"
    def alloc():
        try:
            return ctx.alloc_bodies(n * pitch)
        except m.B3WError as e:                             # a search that ran the device out of memory: plain buffer, and say so
            print(f"bench.py: {e}; falling back to a plain buffer", file=sys.stderr)
            os.environ["B3W_PLACEMENT"] = "plain"
            for attempt in range(6):                        # (a box still releasing the previous process's memory: wait for it)
                ctx.trim()
                m.lib().b3w_bodies_trim()
                try:
                    return ctx.alloc_bodies(n * pitch)
                except m.B3WError as e2:
                    if attempt == 5:
                        raise
                    print(f"bench.py: {e2}; trying again in {2 + attempt} s", file=sys.stderr)
                    time.sleep(2 + attempt)
    t_alloc = time.perf_counter()
    bodies = alloc()
    for attempt in range(4):                                # a box still releasing another process's memory: try again
        if bodies.placement in ("mixed", "interleaved") or os.environ.get("B3W_PLACEMENT") == "plain":   # interleaved: placed, this box's plain buffers were as fast
            break
        cost = ctx.placement_cost()
        if cost["search_timeouts"] or cost["search_gib_walked"] > 1.5 * n * pitch / 2**30:
            break                                           # the search ran out its time or walked far beyond the buffer and found one class: it IS plain here
        bodies.free()
        ctx.trim()                                          # (ring buffers a context keeps from destroyed chains: none here, but say so)
        m.lib().b3w_bodies_trim()                           # hand the pooled pieces back: the next search starts afresh
        time.sleep(1.0 + attempt)
        os.environ["B3W_PLACE_DEBUG"] = "1"                 # say on stderr what the search found
        bodies = alloc()
    d_bodies = bodies                                      # .data_ptr() like a tensor
    place_cost = dict(placement_cost(ctx), placement_alloc_s=round(time.perf_counter() - t_alloc, 3))
    if args.traffic_child is not None:
        # under rocprofv3 --pmc (live_traffic): 2 + 8 launches through each kernel path the parent's autotune may end on — the last 8 of a
        # path are what is counted.  No autotune here; --variant (the parent's, when it was given one) is the only path then.
        paths = {}
        for v in ([args.variant] if args.variant is not None else [0, 200]):
            os.environ["B3W_VARIANT"] = str(v)
            c2 = m.Context(circuit, local_rank)
            try:
                for _ in range(2 + 8):
                    c2.run_device(d_recs.data_ptr(), n, bodies.ptr, pitch, d_pub.data_ptr(), d_status.data_ptr(), stream.cuda_stream)
                torch.cuda.synchronize()
                paths[kernel_path(v)] = 2 if kernel_path(v) == "sweep" or (kernel_path(v) == "fill" and circuit != "compression") else 1
            except m.B3WError:                              # (this circuit has no such path)
                pass
            finally:
                c2.close()
                del os.environ["B3W_VARIANT"]
        print(json.dumps({"traffic_child": True, "launches": 8, "paths": paths}))
        return
    if args.variant is None:
        chosen, best_ms = ctx.autotune_device(d_recs.data_ptr(), n, bodies.ptr, pitch, d_pub.data_ptr(), d_status.data_ptr(),
                                              stream.cuda_stream)
    else:                                                   # (autotune leaves the winner selected in ctx)
        chosen = args.variant
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(2):
        launch(post=args.exchange != "none")                # (the communicator's first collective — connection set-up — stays untimed)
    ev0.record(stream)
    for _ in range(4):
        launch()
    ev1.record(stream)
    ex.finish()
    torch.cuda.synchronize()
    est = torch.tensor([ev0.elapsed_time(ev1) / 4], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(est, op=dist.ReduceOp.MAX)          # every rank must settle on the same launch count
    inner = args.inner if args.inner > 0 else max(1, min(8192, math.ceil(args.timed_ms / (args.steps * max(est.item(), 1e-3)))))
    for _ in range(args.warmup):
        for _ in range(inner):
            launch()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # HIP events on the launch stream around the K * inner launches: average launch duration = region / launches
    # (event records between the launches would cost 7 us each: tools/ubench/launch_gap.py)
    t0 = time.perf_counter()
    ev0.record(stream)
    for i in range(args.steps):
        for j in range(inner):
            launch(post=True if args.exchange == "last" and i == args.steps - 1 and j == inner - 1 else None)
    ev1.record(stream)
    allpub = ex.finish()                                    # every exchange posted above is inside the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    launches = args.steps * inner
    kern_ms = ev0.elapsed_time(ev1) / launches                        # HIP events on the launch stream
    assert int(d_status.abs().sum().item()) == 0, "a witness reported a non-zero status"
    if world > 1 and args.exchange != "none":               # the gathered buffer of the last launch: every rank's outputs
        assert allpub.shape == (world * n, npub) and int((allpub[:, :npub].abs().sum(dim=1) == 0).sum().item()) == 0
        assert torch.equal(allpub[rank * n:(rank + 1) * n], ex.bufs[ex.last]), "this rank's share of the gathered outputs differs from what it sent"
        # ... and the OTHER ranks' shares against what this rank computes itself for their first records (the inputs are a function of the
        # global index): the first run on a real node is the first time a block crosses xGMI — it is checked, not assumed
        k = min(64, n)
        d_k_bodies = torch.empty(k * ctx.body_bytes, dtype=torch.uint8, device=dev)
        d_k_pub = torch.zeros((k, npub), dtype=torch.int32, device=dev)
        d_k_status = torch.zeros(k, dtype=torch.int32, device=dev)
        for q in range(world):
            if q == rank:
                continue
            rq = W.config2_compression(k, first=q * n) if circuit == "compression" else W.config3_nova(k, first=q * n)
            d_rq = torch.from_numpy(rq.view(np.int32)).to(dev)
            ctx.run_device(d_rq.data_ptr(), k, d_k_bodies.data_ptr(), 0, d_k_pub.data_ptr(), d_k_status.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
            assert torch.equal(allpub[q * n:q * n + k, :npub].to(torch.int32), d_k_pub), f"rank {q}'s share of the gathered outputs is not what its inputs give"
    # untimed: every body of the last launch is checked on the device: the rank-1 constraint check
    # Az*Bz = Cz with the constraint system derived from the circuit text where there is one (blake3_compression), and the
    # recompute-and-compare tamper check
    d_mm = torch.full((n,), -1, dtype=torch.int32, device=dev)
    ctx.verify_device(d_bodies.data_ptr(), n, pitch, d_mm.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize()
    assert int(d_mm.abs().sum().item()) == 0, "on-device verification found a body that is not a valid witness"
    verified = "recomputed from own inputs"
    if circuit in m.BUILTIN_R1CS:
        r1cs = m.R1cs(ctx)
        d_viol = torch.full((n,), -1, dtype=torch.int32, device=dev)
        r1cs.check_device(d_bodies.data_ptr(), n, pitch, d_viol.data_ptr(), 0, stream.cuda_stream)
        torch.cuda.synchronize()
        assert int(d_viol.abs().sum().item()) == 0, "a body violates the circuit's rank-1 constraints"
        verified = f"r1cs: 0 of {r1cs.n_constraints} constraints violated by any of the {n} bodies; " + verified
        r1cs.close()

    # untimed, N = 1: (1) the same kernel, same variant, on a plain hipMalloc buffer — what a caller who brings its own buffer gets;
    # (2) the pure-store ceilings of both buffers (b3w_bodies_store_rate: nothing but the stores, in the witness kernels' two shapes):
    # what `achieved` is a fraction of on THIS chip, next to the 8 TB/s of the data sheet.  (The bodies are overwritten: checks are done.)
    plain, ceiling = None, None
    if world == 1:
        # (a baseline leg never fails the bench: the measurement is done; an out-of-memory here — the second buffer is 49 GB for config 3 —
        # or a B3WError must not cost the driver its JSON line)
        d_plain = ctx_plain = None
        try:
            d_plain = torch.empty(n * pitch, dtype=torch.uint8, device=dev)
            # its OWN context and its own autotune ON THIS BUFFER: the launch shape that suits a placed buffer is not the one that suits a plain
            # one, and an integrator who brings a buffer calls b3w_batch_autotune_device on that buffer
            ctx_plain = m.Context(circuit, torch.cuda.current_device())
            if args.variant is None:
                v_plain, _ = ctx_plain.autotune_device(d_recs.data_ptr(), n, d_plain.data_ptr(), pitch, d_pub.data_ptr(), d_status.data_ptr(), stream.cuda_stream)
            else:
                v_plain = args.variant
            ctx_plain.time_device(d_recs.data_ptr(), n, d_plain.data_ptr(), pitch, d_pub.data_ptr(), d_status.data_ptr(), stream.cuda_stream, 5)
            ms_plain = ctx_plain.time_device(d_recs.data_ptr(), n, d_plain.data_ptr(), pitch, d_pub.data_ptr(), d_status.data_ptr(), stream.cuda_stream, 20)
            ach_plain = BYTES_PER_WITNESS[circuit] * n / (ms_plain * 1e-3) / 1e9
            plain = {"kernel_ms": ms_plain, "achieved": ach_plain, "frac": ach_plain / HBM_PEAK_GBS, "launches": 20, "kernel_variant": variant_name(v_plain),
                     "buffer": "torch.empty = hipMalloc, same records; the launch shape autotuned ON THIS BUFFER (b3w_batch_autotune_device, as an integrator "
                               "with its own allocator would), after the timed region"}
            shapes = STORE_SHAPES
            ceiling = {"unit": "GB/s", "what": STORE_SHAPES_WHAT,
                       "placed": {k: ctx.store_rate(d_bodies.data_ptr(), n, pitch, v, 20, stream.cuda_stream) for k, v in shapes.items()},
                       "plain": {k: ctx.store_rate(d_plain.data_ptr(), n, pitch, v, 20, stream.cuda_stream) for k, v in shapes.items()}}
        except Exception as e:
            why = f"{type(e).__name__}: {e}"[:300]
            print(f"bench.py: the untimed plain-buffer / store-ceiling legs failed ({why}); the line goes out without them", file=sys.stderr)
            plain = plain if plain is not None else {"why": why}
            ceiling = {"why": why}
            try:
                torch.cuda.synchronize()
            except Exception:
                pass
        finally:
            if ctx_plain is not None:
                ctx_plain.close()
            del d_plain

    t = torch.tensor([elapsed, kern_ms], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    kern_per_rank = [float(x) for x in gather_strings(dist, world, repr(kern_ms))]     # every rank's own HIP-event average
    # the collective alone (N > 1): 20 all-gathers of the public outputs back to back, nothing else on the device, HIP events
    ex_ms = 0.0
    if world > 1:
        torch.cuda.synchronize()
        dist.barrier()
        ev0.record(stream)
        for _ in range(20):
            ex.next_buffer()
            ex.post()
        ex.finish()
        ev1.record(stream)
        torch.cuda.synchronize()
        ex_ms = ev0.elapsed_time(ev1) / 20
    ex_per_rank = [float(x) for x in gather_strings(dist, world, repr(ex_ms))]
    elapsed, kern_ms = t[0].item(), t[1].item()
    placements = gather_strings(dist, world, bodies.placement)

    if rank == 0:
        total = world * n * launches
        alg_bytes = BYTES_PER_WITNESS[circuit] * n                   # per launch
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        traffic, traffic_source, traffic_parts = None, None, None
        tf = os.path.join(ROOT, "profiles", "traffic_latest.json")   # PMC passes (WRITE_SIZE/FETCH_SIZE), see profiles/README.md
        if measured_traffic is not None:
            if kernel_path(chosen) in measured_traffic:
                measured_traffic = measured_traffic[kernel_path(chosen)]
            else:                                           # (the two-kernel sweep path is not counted live: quote the recorded passes)
                traffic_why_not = f"the counter passes counted the paths {sorted(measured_traffic)}, the timed run is on {kernel_path(chosen)} (variant {chosen})"
                measured_traffic = None
        if measured_traffic is not None:
            traffic = measured_traffic["hbm_bytes_per_launch"]
            traffic_parts = measured_traffic
            traffic_source = ("measured by this invocation: rocprofv3 --pmc WRITE_SIZE and --pmc FETCH_SIZE (separate child passes of 8 launches of this "
                              "batch on a plain buffer, before the timed region), KiB x 1024, FETCH_SIZE x 2 (MI355X_MICROARCH.md, HBM)")
        elif os.path.exists(tf):
            try:
                path = kernel_path(chosen)
                doc = json.load(open(tf))
                for ent in doc.get("entries", []):
                    if ent.get("circuit") == circuit and ent.get("batch") == n and ent.get("path") == path:
                        traffic = ent.get("hbm_bytes_per_launch")
                        traffic_source = (f"NOT measured by this run ({traffic_why_not}): profiles/r{int(doc.get('round', 0)):02d}/{ent.get('tag', circuit)}"
                                          "_pmc_{WRITE,FETCH}_SIZE.csv (separate rocprofv3 --pmc passes of this command)")
            except Exception:
                traffic = None
        cfgname = "config2" if circuit == "compression" else "config3"
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
            "config": {"workload": f"{cfgname}: batch {n} independent {circuit} witnesses per GPU and launch, "
                                   f"{FIELD[circuit]} field, LCG(6429+i) inputs, device-resident inputs and outputs; "
                                   f"one step = {inner} launches over the batch",
                       "circuit": circuit, "batch_per_gpu": n, "launches_per_step": inner, "witnesses_per_step": world * n * inner,
                       "timed_region_s": elapsed, "witness_bytes": ctx.body_bytes, "pitch": pitch,
                       "kernel_variant": variant_name(chosen),
                       "verified_on_device": True, "verification": verified,
                       "placement": placements[0], "placement_per_rank": placements,
                       "exchange": "none" if world == 1 else
                                   f"--exchange {args.exchange}: " + {
                                       "every": f"all_gather of the {n} x {npub} u32 public outputs over {dist.get_world_size()} ranks "
                                                f"({dist.get_backend()}) after every launch, pipelined with the next launch's kernel",
                                       "last": f"all_gather of the {n} x {npub} u32 public outputs over {dist.get_world_size()} ranks "
                                               f"({dist.get_backend()}) after the last timed launch only",
                                       "none": f"no collective inside the timed region ({dist.get_world_size()} ranks, {dist.get_backend()}): kernels only"}[args.exchange],
                       "exchange_mode": args.exchange if world > 1 else "none",
                       "exchange_impl": ("native b3w_comm (" + comm.transport + ")" if comm is not None else "torch.distributed") if world > 1 else "none",
                       "comm_size": (int(m.lib().b3w_comm_size(comm.handle)) if comm is not None else dist.get_world_size()) if world > 1 else 1,
                       "exchange_ms_per_rank": {"public_outputs": ex_per_rank,
                                                "what": "one all-gather of the batch's public outputs with nothing else on the device: HIP events around 20, after the timed region"},
                       "devices_per_rank": devices, **place_cost},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source, "traffic_measured": traffic_parts,
                         "kernel_ms": kern_ms, "kernel_ms_per_rank": kern_per_rank, "kernel_ms_min": min(kern_per_rank),
                         "kernel_ms_max": max(kern_per_rank), "algorithmic_bytes_per_launch": alg_bytes, "launches_timed": launches},
        }
        if plain is not None:
            out["roofline"]["plain"] = plain if "achieved" in plain else None
            out["roofline"]["store_ceiling"] = ceiling if "placed" in ceiling else None
            if "why" in plain or "why" in ceiling:
                out["roofline"]["untimed_legs_failed"] = plain.get("why") or ceiling.get("why")
            if "placed" in ceiling:
                best = max(ceiling["placed"].values())
                out["roofline"]["of_measured_ceiling"] = achieved / best       # of the best store-only shape on the SAME (placed) buffer
                if "achieved" in plain:
                    out["roofline"]["plain_of_measured_ceiling"] = plain["achieved"] / max(ceiling["plain"].values())
        if args.cpu_seconds > 0 and world == 1:
            out["cpu_baseline"] = cpu_baseline(circuit, recs, args.cpu_seconds, args.reference_dir, args.reference_seconds)
        print(json.dumps(out), flush=True)
    if comm is not None:
        comm.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
