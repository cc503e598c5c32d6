"""sliced_scan.py — small batches: one body per wave (variant 1) against the SLICED launch (variant 20 + s: s waves per body,
each storing 1/s of its tiles).  Kernel time by HIP events, outputs compared byte for byte with variant 1."""
import importlib, os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
m = importlib.import_module("hot-proofs-blake3-circom_amd")
st = torch.cuda.current_stream().cuda_stream
circuit = sys.argv[1] if len(sys.argv) > 1 else "compression"
sizes = (1, 4, 16, 64, 128, 256, 512, 1024, 2048, 4096)
slices = (1, 2, 4, 8, 16, 32, 64)
nmax = max(sizes)
os.environ.pop("B3W_VARIANT", None)
ctx0 = m.Context(circuit, 0)
recs = m.workloads.config2_compression(nmax) if circuit == "compression" else m.workloads.config3_nova(nmax)
d_recs = torch.from_numpy(recs.view(np.int32)).cuda()
buf = ctx0.alloc_bodies(nmax * ctx0.body_bytes)
ref = torch.empty((256, ctx0.body_bytes), dtype=torch.uint8, device="cuda")
os.environ["B3W_VARIANT"] = "1"
c1 = m.Context(circuit, 0)
c1.run_device(d_recs.data_ptr(), 256, ref.data_ptr(), 0, 0, 0, st)
torch.cuda.synchronize()
print(f"{circuit}: placement {buf.placement}; M witnesses/s by batch size and slices per body", flush=True)
print("      n " + "".join(f"{'s=' + str(s):>9s}" for s in slices), flush=True)
for n in sizes:
    row = []
    for s in slices:
        os.environ["B3W_VARIANT"] = "1" if s == 1 else str(20 + s)
        ctx = m.Context(circuit, 0)
        for _ in range(3):
            ctx.run_device(d_recs.data_ptr(), n, buf.ptr, 0, 0, 0, st)
        reps = 20 if n <= 1024 else 8
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                ctx.run_device(d_recs.data_ptr(), n, buf.ptr, 0, 0, 0, st)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps)
        k = min(n, 256)                                       # parity with variant 1
        got = torch.zeros((k, ctx.body_bytes), dtype=torch.uint8, device="cuda")
        ctx.run_device(d_recs.data_ptr(), k, got.data_ptr(), 0, 0, 0, st)
        torch.cuda.synchronize()
        assert torch.equal(got, ref[:k]), (n, s)
        row.append((best, n / best / 1e3))
        ctx.close()
    print(f"{n:7d} " + "".join(f"{r[1]:9.2f}" for r in row) + "   us: " + " ".join(f"{r[0] * 1e3:.0f}" for r in row), flush=True)
