"""r1cs_stream_dbg.py [circuit] [n] — where the stream formulation of the constraint check spends its time: the same check with
phases switched off (B3W_R1CS_DBG: 1 no words / rows / verdicts, 2 no pack, 4 or 8 no HBM traffic, 32 no general words) and other
workgroup shapes, each in a child process (the switches are read once per process).  Times only; verdicts are meaningless with a
phase missing."""
import importlib, os, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import numpy as np, torch
    sys.path.insert(0, os.getcwd())
    m = importlib.import_module("hot-proofs-blake3-circom_amd")
    circuit, n = sys.argv[2], int(sys.argv[3])
    s = torch.cuda.current_stream().cuda_stream
    ctx = m.Context(circuit, 0)
    r = m.R1cs(ctx)
    recs = m.workloads.config2_compression(n) if circuit == "compression" else m.workloads.config3_nova(n)
    d_recs = torch.from_numpy(recs.view(np.int32)).cuda()
    bodies = torch.empty((n, ctx.body_bytes), dtype=torch.uint8, device="cuda")
    ctx.run_device(d_recs.data_ptr(), n, bodies.data_ptr(), 0, 0, 0, s)
    viol = torch.zeros(n, dtype=torch.int32, device="cuda")
    for _ in range(2):
        r.check_device(bodies.data_ptr(), n, 0, viol.data_ptr(), 0, s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        r.check_device(bodies.data_ptr(), n, 0, viol.data_ptr(), 0, s)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"{ms:.3f} ms  ({n * ctx.body_bytes / ms / 1e6:.0f} GB/s of body bytes)", flush=True)
    sys.exit(0)
circuit = sys.argv[1] if len(sys.argv) > 1 else "compression"
n = sys.argv[2] if len(sys.argv) > 2 else "4096"
E = lambda **kw: {"B3W_R1CS_" + k.upper(): str(v) for k, v in kw.items()}
# (any B3W_R1CS_DBG value selects the diagnostic instantiation and skips the deferred kernel: compare those lines with "no switch (64)")
for label, env in (("lean pair (GATHER=3)", {"B3W_R1CS_GATHER": "3"}), ("stream, default (8 waves, 2 WG/CU)", {}),
                   ("stream 8 waves, 1 WG/CU", E(wgs=1)), ("stream 16 waves", E(waves=16)),
                   ("diagnostic kernel, no switch (64)", E(dbg=64)), ("no general words (32)", E(dbg=32)),
                   ("fetch + pack, no words / rows / verdicts (1)", E(dbg=1)), ("fetch + barrier (3)", E(dbg=3)),
                   ("all phases, every unit the tile's first body: no HBM traffic (8)", E(dbg=8)), ("loop + barrier, no HBM traffic (11)", E(dbg=11))):
    r = subprocess.run([sys.executable, __file__, "--child", circuit, n], capture_output=True, text=True, env=dict(os.environ, **env), timeout=300)
    print(f"{label:70s} {r.stdout.strip() if r.returncode == 0 else 'FAILED ' + r.stderr[-300:]}", flush=True)
