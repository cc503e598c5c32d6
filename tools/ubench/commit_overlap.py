"""commit_overlap.py — do the witness kernel (HBM-bound) and the commitment from the same records (ALU-bound) overlap
when they run on two streams?  (Both outputs wanted: bodies in HBM and one commitment per witness.)"""
import ctypes, importlib, os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import ec_ref as E
m = importlib.import_module("hot-proofs-blake3-circom_amd")
dev = torch.device("cuda:0")
for circuit, curve, n in (("compression", "bn254_g1", 16384), ("nova_vesta", "vesta", 16384)):
    ctx = m.Context(circuit, 0)
    recs = m.workloads.config2_compression(n) if circuit == "compression" else m.workloads.config3_nova(n)
    key = m.CommitKey(ctx, curve, E.points_to_bytes(E.random_points(curve, ctx.witness_size, seed=b"ovl")), 0, 16)
    d_recs = torch.from_numpy(recs.view(np.int32)).to(dev)
    bodies = ctx.alloc_bodies(n * ctx.body_bytes)
    d_pub = torch.zeros((n, ctx.public_words), dtype=torch.int32, device=dev); d_st = torch.zeros(n, dtype=torch.int32, device=dev)
    d_pts = torch.zeros((n, 64), dtype=torch.uint8, device=dev); d_st2 = torch.zeros(n, dtype=torch.int32, device=dev)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    def gen(s): ctx.run_device(d_recs.data_ptr(), n, bodies.ptr, 0, d_pub.data_ptr(), d_st.data_ptr(), s.cuda_stream)
    def com(s): key.commit_records_device(d_recs.data_ptr(), n, d_pts.data_ptr(), d_st2.data_ptr(), 0, s.cuda_stream)
    def comb(s): key.commit_device(bodies.ptr, n, 0, d_pts.data_ptr(), d_st2.data_ptr(), s.cuda_stream)
    def timed(f, reps=4):
        f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record(torch.cuda.current_stream())
        sa.wait_stream(torch.cuda.current_stream()); sb.wait_stream(torch.cuda.current_stream())
        for _ in range(reps): f()
        torch.cuda.current_stream().wait_stream(sa); torch.cuda.current_stream().wait_stream(sb)
        e1.record(torch.cuda.current_stream()); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    t_gen = timed(lambda: gen(sa)); t_com = timed(lambda: com(sa))
    t_seq = timed(lambda: (gen(sa), comb(sa)))
    t_both = timed(lambda: (gen(sa), com(sb)))
    print(f"{circuit}: bodies {t_gen:.2f} ms, commit from records {t_com:.2f} ms, bodies then commit of the bodies {t_seq:.2f} ms, "
          f"bodies || commit from records on two streams {t_both:.2f} ms ({n / t_both:.0f} k/s)", flush=True)
    key.close(); bodies.free(); ctx.close()
