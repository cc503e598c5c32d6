"""commit_rate_folded.py — the commitment kernel with plain and with FOLDED keys (word slots folded into their bit slots'
generators: fold.py), bodies resident in HBM and straight from the records; the two keys must give the same points."""
import importlib, os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import ec_ref as E
m = importlib.import_module("hot-proofs-blake3-circom_amd")
dev = torch.device("cuda:0"); s = torch.cuda.current_stream().cuda_stream
for circuit, curve, n, window in (("compression", "bn254_g1", 16384, 16), ("compression", "bn254_g1", 4096, 12), ("nova_bn254_o1", "bn254_g1", 8192, 16), ("nova_vesta", "pallas", 8192, 16),
                                  ("nova_vesta", "pallas", 8192, 12)):
    ctx = m.Context(circuit, 0)
    recs = m.workloads.config2_compression(n) if circuit == "compression" else m.workloads.config3_nova(n)
    gens = E.points_to_bytes(E.random_points("vesta" if curve == "pallas" else curve, ctx.witness_size, seed=b"rate"))
    d_recs = torch.from_numpy(recs.view(np.int32)).to(dev)
    bodies = ctx.alloc_bodies(n * ctx.body_bytes)
    ctx.run_device(d_recs.data_ptr(), n, bodies.ptr, 0, 0, 0, s)
    pts = {}
    for fold in (False, True):
        t0 = time.perf_counter()
        key = m.CommitKey(ctx, curve, gens, 0, window, fold=fold)
        tk = time.perf_counter() - t0
        d_pts = torch.zeros((n, 64), dtype=torch.uint8, device=dev); d_st = torch.zeros(n, dtype=torch.int32, device=dev)
        key.commit_device(bodies.ptr, n, 0, d_pts.data_ptr(), d_st.data_ptr(), s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): key.commit_device(bodies.ptr, n, 0, d_pts.data_ptr(), d_st.data_ptr(), s)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        assert int(d_st.abs().sum().item()) == 0
        pts[fold] = d_pts.clone()
        key.commit_records_device(d_recs.data_ptr(), n, d_pts.data_ptr(), d_st.data_ptr(), 0, s)
        e0.record()
        for _ in range(3): key.commit_records_device(d_recs.data_ptr(), n, d_pts.data_ptr(), d_st.data_ptr(), 0, s)
        e1.record(); torch.cuda.synchronize()
        msr = e0.elapsed_time(e1) / 3
        assert torch.equal(d_pts, pts[fold])
        extra = f", {key.fold_stats}" if fold else ""
        print(f"{circuit} on {curve}, {window}-bit windows, {'FOLDED' if fold else 'plain '} key: set-up {tk:.2f} s; {n} bodies in {ms:.2f} ms = "
              f"{n / ms / 1e3:.2f} M/s; from the records {msr:.2f} ms = {n / msr / 1e3:.2f} M/s{extra}", flush=True)
        key.close()
    assert torch.equal(pts[False], pts[True])
    bodies.free(); ctx.close()
