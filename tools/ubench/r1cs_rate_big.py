import importlib, os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
m = importlib.import_module("hot-proofs-blake3-circom_amd")
s = torch.cuda.current_stream().cuda_stream
ctx = m.Context("nova_vesta", 0)
r = m.R1cs(ctx)
for n, placed in ((8192, True), (65536, True), (65536, False)):
    recs = m.workloads.config3_nova(n)
    d_recs = torch.from_numpy(recs.view(np.int32)).cuda()
    if placed:
        buf = ctx.alloc_bodies(n * ctx.body_bytes); ptr = buf.ptr
    else:
        t = torch.empty(n * ctx.body_bytes, dtype=torch.uint8, device="cuda"); ptr = t.data_ptr()
    ctx.run_device(d_recs.data_ptr(), n, ptr, 0, 0, 0, s)
    viol = torch.zeros(n, dtype=torch.int32, device="cuda")
    for _ in range(2): r.check_device(ptr, n, 0, viol.data_ptr(), 0, s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): r.check_device(ptr, n, 0, viol.data_ptr(), 0, s)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(f"n={n} placed={placed}: {ms:.2f} ms = {n / ms / 1e3:.2f} M bodies/s", flush=True)
    # interleaved with generation, as the chain does
    e0.record()
    for _ in range(3):
        ctx.run_device(d_recs.data_ptr(), n, ptr, 0, 0, 0, s)
        r.check_device(ptr, n, 0, viol.data_ptr(), 0, s)
    e1.record(); torch.cuda.synchronize()
    ms2 = e0.elapsed_time(e1) / 3
    print(f"   generate + check: {ms2:.2f} ms -> check share {ms2 - n / 9.3e3:.2f} ms", flush=True)
    if placed: buf.free()
    else: del t
