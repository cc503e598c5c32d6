"""r1cs_rate.py — bodies/s of the constraint check for every circuit with a derived system (4 096 / 16 384 valid bodies)."""
import importlib, os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
m = importlib.import_module("hot-proofs-blake3-circom_amd")
s = torch.cuda.current_stream().cuda_stream
for circuit in ("compression", "nova_bn254_o1", "nova_bn254", "nova_vesta"):
    ctx = m.Context(circuit, 0)
    r = m.R1cs(ctx)
    for n in (4096, 16384):
        recs = m.workloads.config2_compression(n) if circuit == "compression" else m.workloads.config3_nova(n)
        d_recs = torch.from_numpy(recs.view(np.int32)).cuda()
        buf = ctx.alloc_bodies(n * ctx.body_bytes)
        ctx.run_device(d_recs.data_ptr(), n, buf.ptr, 0, 0, 0, s)
        viol = torch.zeros(n, dtype=torch.int32, device="cuda")
        for _ in range(2):
            r.check_device(buf.ptr, n, 0, viol.data_ptr(), 0, s)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            r.check_device(buf.ptr, n, 0, viol.data_ptr(), 0, s)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        assert int(viol.abs().sum().item()) == 0
        print(f"{circuit:14s} n={n:6d} constraints {r.n_constraints} terms {r.n_terms}: {ms:.2f} ms = {n / ms / 1e3:.2f} M bodies/s", flush=True)
        buf.free()
    r.close(); ctx.close()
