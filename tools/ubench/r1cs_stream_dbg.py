"""r1cs_stream_dbg.py [circuit] [n] — where the stream formulation of the constraint check spends its time: the same check with
phases switched off (B3W_R1CS_DBG: 1 no rows, 2 no pack, 4 no DMA, 16 no second row pass, 32 no general words), ring depths and grid sizes, each in a child process (the
switches are read once per process).  Times only; verdicts are meaningless with a phase missing."""
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
for label, env in (("lean pair (GATHER=3)", {"B3W_R1CS_GATHER": "3"}), ("stream, default shape", {}),
                   ("stream 8 waves, 1 buffer, 2 WG/CU", E(waves=8, nbuf=1, wgs=2)), ("... tile loads without nt (64)", E(waves=8, nbuf=1, wgs=2, dbg=64)),
                   ("... outside-wire loads nt too (128)", E(waves=8, nbuf=1, wgs=2, dbg=128)),
                   ("stream 8 waves, 1 buffer, 1 WG/CU", E(waves=8, nbuf=1, wgs=1)), ("stream 8 waves, 2 buffers, 1 WG/CU", E(waves=8, nbuf=2, wgs=1)),
                   ("stream 16 waves, 2 buffers", E(waves=16, nbuf=2)), ("stream 16 waves, 3 buffers", E(waves=16, nbuf=3)),
                   ("default: no rows (1)", E(dbg=1)), ("default: DMA + barriers (3)", E(dbg=3)),
                   ("default: rows on valid data, no DMA (8)", E(dbg=8)), ("default: loop + barriers (7)", E(dbg=7))):
    r = subprocess.run([sys.executable, __file__, "--child", circuit, n], capture_output=True, text=True, env=dict(os.environ, **env), timeout=300)
    print(f"{label:40s} {r.stdout.strip() if r.returncode == 0 else 'FAILED ' + r.stderr[-300:]}", flush=True)
