"""r1cs_walk_grid.py <circuit> — the walk kernel with 1, 2, 3 workgroups per CU's worth of persistent workgroups (B3W_R1CS_GRID), on
12 288 bodies (a whole number of bodies per workgroup for each): does more data in flight buy anything?"""
import os, subprocess, sys
script = r'''
import importlib, os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
m = importlib.import_module("hot-proofs-blake3-circom_amd")
circuit = sys.argv[1]; n = 12288
s = torch.cuda.current_stream().cuda_stream
ctx = m.Context(circuit, 0)
r = m.R1cs(ctx)
recs = m.workloads.config2_compression(n) if circuit == "compression" else m.workloads.config3_nova(n)
d_recs = torch.from_numpy(recs.view(np.int32)).cuda()
bodies = torch.empty((n, ctx.body_bytes), dtype=torch.uint8, device="cuda")
ctx.run_device(d_recs.data_ptr(), n, bodies.data_ptr(), 0, 0, 0, s)
viol = torch.zeros(n, dtype=torch.int32, device="cuda")
for _ in range(2): r.check_device(bodies.data_ptr(), n, 0, viol.data_ptr(), 0, s)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): r.check_device(bodies.data_ptr(), n, 0, viol.data_ptr(), 0, s)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
assert int(viol.abs().sum().item()) == 0
print(f"{ms:.3f} ms = {n / ms / 1e3:.2f} M bodies/s = {n * ctx.body_bytes / ms / 1e9:.2f} TB/s of body bytes")
'''
for grid in (256, 384, 512, 768, 1024):
    r = subprocess.run([sys.executable, "-c", script, sys.argv[1]], capture_output=True, text=True, env=dict(os.environ, B3W_R1CS_GRID=str(grid)))
    print(f"{sys.argv[1]} grid {grid:5d}: {r.stdout.strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else r.stderr[-300:]}", flush=True)
