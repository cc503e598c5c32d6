"""Read-side rate of the on-device consumer (b3w_batch_verify_device), on a plain and on a placed body buffer."""
import importlib, sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
m = importlib.import_module("hot-proofs-blake3-circom_amd")
dev = torch.device("cuda:0")
for circuit, n in (("compression", 4096), ("nova_vesta", 16384)):
    ctx = m.Context(circuit, 0)
    W = m.workloads
    recs = W.config2_compression(n) if circuit == "compression" else W.config3_nova(n)
    d_recs = torch.from_numpy(recs.view(np.int32)).to(dev)
    for placement in ("plain", "mixed"):
        if placement == "plain":
            d_bodies = torch.empty((n, ctx.body_bytes), dtype=torch.uint8, device=dev)
        else:
            d_bodies = ctx.alloc_bodies(n * ctx.body_bytes)
        d_st = torch.zeros(n, dtype=torch.int32, device=dev); d_mm = torch.zeros(n, dtype=torch.int32, device=dev)
        s = torch.cuda.current_stream()
        ctx.run_device(d_recs.data_ptr(), n, d_bodies.data_ptr(), 0, 0, d_st.data_ptr(), s.cuda_stream)
        for _ in range(3): ctx.verify_device(d_bodies.data_ptr(), n, 0, d_mm.data_ptr(), s.cuda_stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(20): ctx.verify_device(d_bodies.data_ptr(), n, 0, d_mm.data_ptr(), s.cuda_stream)
        e1.record(s); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        assert int(d_mm.abs().sum().item()) == 0
        print(f"{circuit} [{placement}]: verify {n} bodies in {ms:.3f} ms = {n/ms/1e3:.2f} M bodies/s, {n*ctx.body_bytes/ms/1e6:.0f} GB/s read")
        if placement == "mixed":
            d_bodies.free()
    ctx.close()
