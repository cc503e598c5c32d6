"""small_batch_footprint.py — are the small-batch rates Infinity Cache rates?  A batch of n bodies is launched 32 times, once over
the SAME n-body window and once walking over `k` different windows of a large buffer (footprint k * n bodies): if the 256 MiB
die-level cache absorbs the stores of a repeated small batch, the walking rate is the one a caller with fresh buffers sees."""
import importlib, os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
m = importlib.import_module("hot-proofs-blake3-circom_amd")
st = torch.cuda.current_stream().cuda_stream
ctx = m.Context("compression", 0)
nmax = 16384
recs = m.workloads.config2_compression(nmax)
d_recs = torch.from_numpy(recs.view(np.int32)).cuda()
buf = ctx.alloc_bodies(nmax * ctx.body_bytes)
print(f"placement {buf.placement}; body {ctx.body_bytes} B; 256 MiB = {256 * 2**20 // ctx.body_bytes} bodies", flush=True)
for n in (16, 64, 128, 256, 320, 384, 448, 512, 1024, 2048):
    out = []
    for windows in (1, max(1, nmax // n)):
        def run():
            for i in range(32):
                w = i % windows
                ctx.run_device(d_recs.data_ptr() + w * n * 112, n, buf.ptr + w * n * ctx.body_bytes, 0, 0, 0, st)
        run(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 32)
        out.append((windows, n / best / 1e3, n * ctx.body_bytes / best / 1e9))
    print(f"n={n:5d} ({n * ctx.body_bytes / 2**20:7.1f} MiB): same window {out[0][1]:.2f} M/s = {out[0][2]:.2f} TB/s | "
          f"{out[1][0]} windows {out[1][1]:.2f} M/s = {out[1][2]:.2f} TB/s", flush=True)
