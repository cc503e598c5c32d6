"""fill_fuzz.py — the fill-ordered path against the body-stream kernel on random shapes: batch sizes 1 .. 6 000, pitches (contiguous, padded by
32 .. 200 000 bytes), buffer starts at any 32-byte step of a region, rejected nova steps sprinkled in; every byte of the span compared
(bodies, gaps and margins), outputs and status too.  python tools/ubench/fill_fuzz.py [seconds=90] [seed=1]"""
import importlib, os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
m = importlib.import_module("hot-proofs-blake3-circom_amd")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 90.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
st = torch.cuda.current_stream().cuda_stream
dev = torch.device("cuda:0")
ctxs = {}
for circuit in ("compression", "nova_vesta", "nova_bn254"):
    for v in ("0", "200", "201"):
        os.environ["B3W_VARIANT"] = v
        ctxs[circuit, v] = m.Context(circuit, 0)
os.environ.pop("B3W_VARIANT")
t0, cases, worst = time.time(), 0, 0
while time.time() - t0 < budget:
    circuit = ("compression", "nova_vesta", "nova_bn254")[int(rng.integers(3))]
    n = int(rng.choice([1, 2, 3, int(rng.integers(4, 200)), int(rng.integers(200, 1500)), int(rng.integers(1500, 6000))]))
    ref, fill = ctxs[circuit, "0"], ctxs[circuit, str(rng.choice(["200", "201"]))]
    body = ref.body_bytes
    pad = int(rng.choice([0, 0, 32, 96, 4096, 32 * int(rng.integers(1, 6000))]))
    pitch = body + pad
    skew = 32 * int(rng.integers(0, 4096))
    first = int(rng.integers(0, 1 << 20))
    recs = (m.workloads.config2_compression(n, first=first) if circuit == "compression" else m.workloads.config3_nova(n, first=first)).copy()
    if circuit != "compression":
        for i in rng.integers(0, n, size=max(1, n // 50)):
            recs[i, 14] = recs[i, 12]                         # depth = leaf_depth: rejected
    d_recs = torch.from_numpy(recs.view(np.int32)).to(dev)
    span = n * pitch + (1 << 18)
    bufs, outs = [], []
    for ctx in (ref, fill):
        buf = torch.full((span + (1 << 17),), 0x5A, dtype=torch.uint8, device=dev)
        lo = (1 << 17) - (buf.data_ptr() % (1 << 17)) + skew
        lo %= (1 << 17)
        npub = 16 if circuit == "compression" else 15
        d_pub = torch.zeros((n, npub), dtype=torch.int32, device=dev)
        d_st = torch.full((n,), -7, dtype=torch.int32, device=dev)
        ctx.run_device(d_recs.data_ptr(), n, buf.data_ptr() + lo, pitch, d_pub.data_ptr(), d_st.data_ptr(), st)
        bufs.append(buf[lo:lo + n * pitch + 4096]); outs.append((d_pub, d_st))
    torch.cuda.synchronize()
    ok = outs[0][1] == 0
    assert torch.equal(outs[0][1], outs[1][1]), (circuit, n, pitch, skew, "status")
    assert torch.equal(outs[0][0][ok], outs[1][0][ok]), (circuit, n, pitch, skew, "outputs")
    assert torch.equal(bufs[0], bufs[1]), (circuit, n, pitch, skew, "bytes")
    cases += 1; worst = max(worst, n)
    del bufs, outs
print(f"fill_fuzz: {cases} random shapes in {time.time() - t0:.0f} s (largest batch {worst}): the fill-ordered path (variants 200, 201) byte-equal to the body-stream kernel, bodies, gaps, margins, outputs, status")
