"""r1cs_stream_dbg.py [circuit] [n] — where the stream formulation of the constraint check spends its time: the same check with
phases switched off (B3W_R1CS_DBG: 1 no rows, 2 no pack, 4 no DMA), ring depths and grid sizes, each in a child process (the
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
for label, env in (("lean pair (GATHER=3)", {"B3W_R1CS_GATHER": "3"}), ("stream, 3 buffers", {}), ("stream, 2 buffers", {"B3W_R1CS_NBUF": "2"}),
                   ("  no rows", {"B3W_R1CS_DBG": "1"}), ("  no rows, no pack (DMA + barriers)", {"B3W_R1CS_DBG": "3"}),
                   ("  no DMA (pack + rows)", {"B3W_R1CS_DBG": "4"}), ("  no DMA, no pack (rows)", {"B3W_R1CS_DBG": "6"}),
                   ("  nothing (loop + barriers)", {"B3W_R1CS_DBG": "7"}),
                   ("  rows on valid data, staged once (8)", {"B3W_R1CS_DBG": "8"}), ("  ... without the second row pass (24)", {"B3W_R1CS_DBG": "24"}),
                   ("  ... booleanity rows only (40)", {"B3W_R1CS_DBG": "40"}), ("  ... booleanity only, no second pass (56)", {"B3W_R1CS_DBG": "56"}),
                   ("  all but the second row pass (16)", {"B3W_R1CS_DBG": "16"}), ("  all but the general rows (32)", {"B3W_R1CS_DBG": "32"})):
    r = subprocess.run([sys.executable, __file__, "--child", circuit, n], capture_output=True, text=True, env=dict(os.environ, **env), timeout=300)
    print(f"{label:40s} {r.stdout.strip() if r.returncode == 0 else 'FAILED ' + r.stderr[-300:]}", flush=True)
