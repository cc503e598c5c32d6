"""variants_placed.py — kernel variants (bodies per wave) and pitches on a placed (two-class) body buffer."""
import importlib, os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
m = importlib.import_module("hot-proofs-blake3-circom_amd")
dev = torch.device("cuda:0")
s = torch.cuda.current_stream().cuda_stream
def run(circuit, n, variants, pitches):
    recs = m.workloads.config2_compression(n) if circuit == "compression" else m.workloads.config3_nova(n)
    d_recs = torch.from_numpy(recs.view(np.int32)).to(dev)
    base = m.Context(circuit, 0)
    d_pub = torch.zeros((n, base.public_words), dtype=torch.int32, device=dev); d_st = torch.zeros(n, dtype=torch.int32, device=dev)
    buf = base.alloc_bodies(n * max(pitches) + 4096)
    per = base.body_bytes + 4 * recs.shape[1]
    print(circuit, n, "placement", buf.placement, flush=True)
    for v in variants:
        os.environ["B3W_VARIANT"] = str(v)
        ctx = m.Context(circuit, 0)
        for pitch in pitches:
            for off in (0, 32):
                try:
                    for _ in range(3): ctx.run_device(d_recs.data_ptr(), n, buf.ptr + off, pitch, d_pub.data_ptr(), d_st.data_ptr(), s)
                    ms = min(ctx.time_device(d_recs.data_ptr(), n, buf.ptr + off, pitch, d_pub.data_ptr(), d_st.data_ptr(), s, 10) for _ in range(3))
                    print(f"  variant {v:3d} pitch {pitch} off {off:2d}: {ms:.4f} ms {n * per / ms / 1e6:6.0f} GB/s", flush=True)
                except Exception as e:
                    print(f"  variant {v} pitch {pitch}: {e}")
        ctx.close()
    del os.environ["B3W_VARIANT"]
    buf.free(); base.close()
run("compression", 4096, (0, 1, 2, 3, 7, 100), (770976, 771072, 774144))
run("nova_vesta", 16384, (0, 1, 2, 100), (745312, 745472))
run("nova_bn254_o1", 16384, (0, 1), (787648,))
