"""occupancy.py — does limiting resident workgroups per CU (dynamic LDS padding) help large batches?"""
import importlib, os, subprocess, sys
if len(sys.argv) > 1:
    import numpy as np, torch
    sys.path.insert(0, os.getcwd())
    m = importlib.import_module("hot-proofs-blake3-circom_amd")
    circuit, n = sys.argv[1], int(sys.argv[2])
    dev = torch.device("cuda:0"); s = torch.cuda.current_stream().cuda_stream
    recs = m.workloads.config2_compression(n) if circuit == "compression" else m.workloads.config3_nova(n)
    d_recs = torch.from_numpy(recs.view(np.int32)).to(dev)
    ctx = m.Context(circuit, 0)
    d_pub = torch.zeros((n, ctx.public_words), dtype=torch.int32, device=dev); d_st = torch.zeros(n, dtype=torch.int32, device=dev)
    buf = ctx.alloc_bodies(n * ctx.body_bytes)
    per = ctx.body_bytes + 4 * recs.shape[1]
    for _ in range(3): ctx.run_device(d_recs.data_ptr(), n, buf.ptr, 0, d_pub.data_ptr(), d_st.data_ptr(), s)
    ms = min(ctx.time_device(d_recs.data_ptr(), n, buf.ptr, 0, d_pub.data_ptr(), d_st.data_ptr(), s, 10) for _ in range(3))
    print(f"{circuit} n={n} variant={os.environ.get('B3W_VARIANT','0')} pad={os.environ.get('B3W_LDS_PAD','0')}: {ms:.4f} ms {n * per / ms / 1e6:6.0f} GB/s ({buf.placement})", flush=True)
else:
    for circuit, n, variants in (("compression", 32768, ("0", "3")), ("nova_vesta", 32768, ("0",))):
        for v in variants:
            for pad in (0, 8192, 16384, 24576, 32768, 49152):
                env = dict(os.environ, B3W_LDS_PAD=str(pad), B3W_VARIANT=v)
                subprocess.run([sys.executable, __file__, circuit, str(n)], env=env)
