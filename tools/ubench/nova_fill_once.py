"""nova_fill_once.py — 20 launches of the nova fill-ordered path (variant 200) on a one-class buffer, for rocprofv3 --kernel-trace --stats:
the fill kernel's and the wide-slot launch's durations side by side.  python tools/ubench/nova_fill_once.py [n=16384]"""
import importlib, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
m = importlib.import_module("hot-proofs-blake3-circom_amd")
st = torch.cuda.current_stream().cuda_stream
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
recs = m.workloads.config3_nova(n)
d_recs = torch.from_numpy(recs.view(np.int32)).cuda()
os.environ["B3W_VARIANT"] = "200"
ctx = m.Context("nova_vesta", 0)
os.environ["B3W_PLACEMENT"] = "single"
buf = ctx.alloc_bodies(n * ctx.body_bytes)
for _ in range(20):
    ctx.run_device(d_recs.data_ptr(), n, buf.ptr, 0, 0, 0, st)
torch.cuda.synchronize()
