"""commit_rate.py — throughput of the commitment kernel (b3w_batch_commit_device) on resident bodies."""
import ctypes, importlib, os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import ec_ref as E
m = importlib.import_module("hot-proofs-blake3-circom_amd")
L = m.lib()
dev = torch.device("cuda:0"); s = torch.cuda.current_stream().cuda_stream
for circuit, curve, n, window in (("compression", "bn254_g1", 4096, 12), ("compression", "bn254_g1", 16384, 12), ("nova_vesta", "vesta", 8192, 12),
                                  ("compression", "bn254_g1", 4096, 16), ("compression", "bn254_g1", 16384, 16), ("nova_vesta", "vesta", 8192, 16)):
    ctx = m.Context(circuit, 0)
    recs = m.workloads.config2_compression(n) if circuit == "compression" else m.workloads.config3_nova(n)
    gens = E.random_points(curve, ctx.witness_size, seed=b"rate")
    key = ctypes.c_void_p()
    t0 = time.perf_counter()
    assert L.b3w_commit_key_create_ex(ctx.handle, E.CURVE_ID[curve], 0, E.points_to_bytes(gens), window, ctypes.byref(key)) == 0
    tk = time.perf_counter() - t0
    d_recs = torch.from_numpy(recs.view(np.int32)).to(dev)
    bodies = ctx.alloc_bodies(n * ctx.body_bytes)
    d_pts = torch.zeros((n, 64), dtype=torch.uint8, device=dev); d_st = torch.zeros(n, dtype=torch.int32, device=dev)
    ctx.run_device(d_recs.data_ptr(), n, bodies.ptr, 0, 0, 0, s)
    L.b3w_batch_commit_device(ctx.handle, key, bodies.ptr, n, 0, d_pts.data_ptr(), d_st.data_ptr(), s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): L.b3w_batch_commit_device(ctx.handle, key, bodies.ptr, n, 0, d_pts.data_ptr(), d_st.data_ptr(), s)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    assert int(d_st.abs().sum().item()) == 0
    e0.record()
    for _ in range(3): L.b3w_commit_records_device(ctx.handle, key, d_recs.data_ptr(), n, d_pts.data_ptr(), None, d_st.data_ptr(), s)
    e1.record(); torch.cuda.synchronize()
    msr = e0.elapsed_time(e1) / 3
    assert int(d_st.abs().sum().item()) == 0
    print(f"{circuit} on {curve}, {window}-bit windows: key set-up {tk*1e3:.0f} ms; commit {n} bodies in {ms:.2f} ms = {n/ms:.1f} k witnesses/s; from the records (no bodies) {msr:.2f} ms = {n/msr:.1f} k/s", flush=True)
    L.b3w_commit_key_destroy(key); bodies.free(); ctx.close()
