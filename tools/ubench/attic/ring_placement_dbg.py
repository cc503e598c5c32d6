"""ring_placement_dbg.py — what b3w_bodies_alloc decides for the chained pass's ring buffers (12.2 GB each, nova) and why."""
import importlib, os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
os.environ["B3W_PLACE_DEBUG"] = "1"
m = importlib.import_module("hot-proofs-blake3-circom_amd")
ctx = m.Context("nova_vesta", 0)
bufs = [ctx.alloc_bodies(16384 * ctx.body_bytes) for _ in range(2)]
print([b.placement for b in bufs], ctx.bodies_stats(), flush=True)
n = 16384
recs = m.workloads.config3_nova(n)
d_recs = torch.from_numpy(recs.view(np.int32)).cuda()
d_st = torch.zeros(n, dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
plain = torch.empty(n * ctx.body_bytes, dtype=torch.uint8, device="cuda")
for name, ptr in [("plain", plain.data_ptr()), ("ring0", bufs[0].ptr), ("ring1", bufs[1].ptr)]:
    for _ in range(2):
        ctx.run_device(d_recs.data_ptr(), n, ptr, 0, 0, d_st.data_ptr(), s)
    ms = ctx.time_device(d_recs.data_ptr(), n, ptr, 0, 0, d_st.data_ptr(), s, 5)
    print(name, f"{n * ctx.body_bytes / ms / 1e6:.0f} GB/s", flush=True)
