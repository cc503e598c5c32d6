"""r1cs_walk_scaling.py <circuit> — ms per check against the batch size (plain torch buffer): what a launch costs beyond its bodies"""
import importlib, os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
m = importlib.import_module("hot-proofs-blake3-circom_amd")
circuit = sys.argv[1]
s = torch.cuda.current_stream().cuda_stream
ctx = m.Context(circuit, 0)
r = m.R1cs(ctx)
nmax = 16384
recs = m.workloads.config2_compression(nmax) if circuit == "compression" else m.workloads.config3_nova(nmax)
d_recs = torch.from_numpy(recs.view(np.int32)).cuda()
bodies = torch.empty((nmax, ctx.body_bytes), dtype=torch.uint8, device="cuda")
ctx.run_device(d_recs.data_ptr(), nmax, bodies.data_ptr(), 0, 0, 0, s)
viol = torch.zeros(nmax, dtype=torch.int32, device="cuda")
for n in (512, 1024, 2048, 4096, 8192, 16384):
    for _ in range(3): r.check_device(bodies.data_ptr(), n, 0, viol.data_ptr(), 0, s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): r.check_device(bodies.data_ptr(), n, 0, viol.data_ptr(), 0, s)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{circuit} n={n:6d}: {ms * 1e3:8.1f} us per check = {ms * 1e3 / n * 4096:7.1f} us per 4 096 bodies = {n * ctx.body_bytes / ms / 1e9:5.2f} TB/s", flush=True)
assert int(viol.abs().sum().item()) == 0
