"""fill_ramp.py — the fill-ordered kernel's fixed cost: kernel time by batch size on a one-class buffer (HIP events around 5 back-to-back
launches, best of 3), compression, variants 200 and 0.  T(n) = F + n t."""
import importlib, os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
m = importlib.import_module("hot-proofs-blake3-circom_amd")
st = torch.cuda.current_stream().cuda_stream
nmax = 32768
recs = m.workloads.config2_compression(nmax)
d_recs = torch.from_numpy(recs.view(np.int32)).cuda()
for variant in ("200", "0"):
    os.environ["B3W_VARIANT"] = variant
    ctx = m.Context("compression", 0)
    os.environ["B3W_PLACEMENT"] = "single"
    buf = ctx.alloc_bodies(nmax * ctx.body_bytes)
    os.environ.pop("B3W_PLACEMENT")
    rows = []
    for n in (1024, 2048, 3072, 4096, 8192, 16384, 32768):
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
        rows.append((n, round(best * 1e3, 1), round(n * 771088 / best / 1e9, 3)))
    print("variant", variant, "(n, us per launch, TB/s):", rows, flush=True)
    buf.free(); ctx.close()
