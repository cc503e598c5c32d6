import importlib, os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
m = importlib.import_module("hot-proofs-blake3-circom_amd")
dev = torch.device("cuda:0")
ctx = m.Context("compression", 0)
r1cs = m.R1cs(ctx)
n = 256
recs = torch.from_numpy(m.workloads.config2_compression(n).view(np.int32)).to(dev)
bodies = torch.zeros((n, ctx.body_bytes), dtype=torch.uint8, device=dev)
st = torch.full((n,), -1, dtype=torch.int32, device=dev)
mm = torch.full((n,), -1, dtype=torch.int32, device=dev)
viol = torch.full((n,), -1, dtype=torch.int32, device=dev)
side = torch.cuda.Stream()
def enq(s, what):
    if "w" in what: ctx.run_device(recs.data_ptr(), n, bodies.data_ptr(), 0, 0, st.data_ptr(), s)
    if "v" in what: ctx.verify_device(bodies.data_ptr(), n, 0, mm.data_ptr(), s)
    if "r" in what: r1cs.check_device(bodies.data_ptr(), n, 0, viol.data_ptr(), 0, s)
for what in ("w", "wv", "wr", "wvr"):
    with torch.cuda.stream(side):
        enq(side.cuda_stream, what)
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, stream=side):
            enq(torch.cuda.current_stream().cuda_stream, what)
        st.fill_(-1); mm.fill_(-1); viol.fill_(-1)
        g.replay(); torch.cuda.synchronize()
        print(what, "st", int(st.abs().sum()), "mm", int(mm.abs().sum()), "viol", int(viol.abs().sum()), flush=True)
    except Exception as e:
        print(what, "capture failed:", str(e)[:300], flush=True)
