import importlib, sys, os, ctypes, numpy as np, torch
sys.path.insert(0, os.getcwd())
os.environ["B3W_VARIANT"] = "100"
m = importlib.import_module("hot-proofs-blake3-circom_amd")
ctx = m.Context("compression", 0)
L = m.lib()
n = 4096
recs = m.workloads.config2_compression(n)
dev = torch.device("cuda:0")
d_recs = torch.from_numpy(recs.view(np.int32)).to(dev)
d_pub = torch.zeros((n,16), dtype=torch.int32, device=dev); d_st = torch.zeros(n, dtype=torch.int32, device=dev)
buf = torch.empty(n*ctx.body_bytes, dtype=torch.uint8, device=dev)
s = torch.cuda.current_stream().cuda_stream
for dbg in (0, 1, 2, 3, 0):
    L.b3w_set_sweep_dbg(ctypes.c_uint32(dbg))
    for _ in range(3): ctx.run_device(d_recs.data_ptr(), n, buf.data_ptr(), 0, d_pub.data_ptr(), d_st.data_ptr(), s)
    ms = ctx.time_device(d_recs.data_ptr(), n, buf.data_ptr(), 0, d_pub.data_ptr(), d_st.data_ptr(), s, 20)
    print(f"dbg={dbg} (1=no gathers, 2=no table): {ms:.4f} ms  {n*771088/ms/1e6:.0f} GB/s", flush=True)
