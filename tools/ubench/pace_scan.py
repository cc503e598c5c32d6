"""pace_scan.py — the fill-ordered kernel (variant 200) on a ONE-class buffer by B3W_FILL_PACE: s_sleep 1 per unit and storing wave, 0 .. 8
(csrc/b3w_kernels.hip, \"PACE\" at the launch): where the optimum lies for the compression circuit and for the nova O2 builds."""
import importlib, os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
m = importlib.import_module("hot-proofs-blake3-circom_amd")
st = torch.cuda.current_stream().cuda_stream
for circuit, n in (("compression", 4096), ("nova_vesta", 8192)):
    recs = m.workloads.config2_compression(n) if circuit == "compression" else m.workloads.config3_nova(n)
    d_recs = torch.from_numpy(recs.view(np.int32)).cuda()
    os.environ["B3W_VARIANT"] = "200"
    ctx = m.Context(circuit, 0)
    os.environ["B3W_PLACEMENT"] = "single"
    buf = ctx.alloc_bodies(n * ctx.body_bytes)
    os.environ.pop("B3W_PLACEMENT")
    algo = n * (ctx.body_bytes + (112 if circuit == "compression" else 128))
    row = []
    for pace in [int(a) for a in sys.argv[1:]] or (0, 1, 2, 3, 4, 5, 6, 8):
        os.environ["B3W_FILL_PACE"] = str(pace)
        for _ in range(2):
            ctx.run_device(d_recs.data_ptr(), n, buf.ptr, 0, 0, 0, st)
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ctx.run_device(d_recs.data_ptr(), n, buf.ptr, 0, 0, 0, st)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 5)
        row.append((pace, round(algo / best / 1e9, 3)))
    print(circuit, "1-class TB/s by pace (s_sleep 1 per unit and storing wave):", row, flush=True)
    buf.free(); ctx.close()
