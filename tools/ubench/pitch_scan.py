import importlib, os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
m = importlib.import_module("hot-proofs-blake3-circom_amd")
st = torch.cuda.current_stream().cuda_stream
n = 4096
recs = m.workloads.config2_compression(n)
d_recs = torch.from_numpy(recs.view(np.int32)).cuda()
os.environ["B3W_VARIANT"] = "0"
ctx = m.Context("compression", 0)
body = ctx.body_bytes
placed = ctx.alloc_bodies(n * (body + 4096))
print("placement", placed.placement)
for rnd in range(2):
    row = []
    for pad in (0, 32, 64, 96, 128, 224, 96 + 1024):
        pitch = body + pad
        for _ in range(3):
            ctx.run_device(d_recs.data_ptr(), n, placed.ptr, pitch, 0, 0, st)
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ctx.run_device(d_recs.data_ptr(), n, placed.ptr, pitch, 0, 0, st)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20)
        row.append((pad, pitch % 128, round(n * 771088 / best / 1e9, 3)))
    print("variant 0, placed: (pad, pitch % 128, TB/s)", row, flush=True)
