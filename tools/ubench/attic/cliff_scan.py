"""cliff_scan.py — where exactly does the one-body-per-wave kernel fall off between 384 and 512 witnesses (batch curve:
384 -> 5.7 M/s, 512 -> 3.6 M/s)?  Times n = 256 ... 1100 in steps of 16, compression and nova."""
import importlib, os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
os.environ["B3W_VARIANT"] = "1"
os.environ["B3W_PLACE_CHECK"] = "0"
m = importlib.import_module("hot-proofs-blake3-circom_amd")
s = torch.cuda.current_stream().cuda_stream
for circuit in ("compression", "nova_vesta"):
    ctx = m.Context(circuit, 0)
    nmax = 1100
    recs = m.workloads.config2_compression(nmax) if circuit == "compression" else m.workloads.config3_nova(nmax)
    d_recs = torch.from_numpy(recs.view(np.int32)).cuda()
    d_st = torch.zeros(nmax, dtype=torch.int32, device="cuda")
    buf = ctx.alloc_bodies(nmax * ctx.body_bytes)
    line = []
    for n in list(range(256, 1101, 16)):
        for _ in range(2):
            ctx.run_device(d_recs.data_ptr(), n, buf.ptr, 0, 0, d_st.data_ptr(), s)
        ms = min(ctx.time_device(d_recs.data_ptr(), n, buf.ptr, 0, 0, d_st.data_ptr(), s, 30) for _ in range(3))
        line.append(f"{n}:{ms * 1e3:.0f}")
    print(circuit, "us per launch:", " ".join(line), flush=True)
    buf.free(); ctx.close()
