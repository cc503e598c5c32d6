"""r1cs_profile_target.py <circuit> [n] — what tools/profile_r1cs.sh runs under rocprofv3: one batch of valid witnesses, then
3 + 10 constraint checks of it (the first three are warm-up: the first launch of a process runs 15 % longer — cold instruction
cache and program tables — and tools/profile_r1cs_collect.py averages the LAST ten launches of each kernel)."""
import importlib, os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
m = importlib.import_module("hot-proofs-blake3-circom_amd")
circuit = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
s = torch.cuda.current_stream().cuda_stream
ctx = m.Context(circuit, 0)
r = m.R1cs(ctx)
recs = m.workloads.config2_compression(n) if circuit == "compression" else m.workloads.config3_nova(n)
d_recs = torch.from_numpy(recs.view(np.int32)).cuda()
bodies = torch.empty((n, ctx.body_bytes), dtype=torch.uint8, device="cuda")
ctx.run_device(d_recs.data_ptr(), n, bodies.data_ptr(), 0, 0, 0, s)
viol = torch.zeros(n, dtype=torch.int32, device="cuda")
for _ in range(13):
    r.check_device(bodies.data_ptr(), n, 0, viol.data_ptr(), 0, s)
torch.cuda.synchronize()
assert int(viol.abs().sum().item()) == 0
print(f"{circuit}: {n} bodies x {ctx.body_bytes} B = {n * ctx.body_bytes} body bytes per check")
