"""variant_scan.py — every launch shape of the witness kernels on three kinds of body buffer: a plain hipMalloc one (torch), a ONE-class
one made on purpose (B3W_PLACEMENT=single: what a plain buffer is on an unlucky day) and a placed one.  Kernel time by HIP events
(b3w_batch_time_device), TB/s of algorithmic bytes; every shape's first 64 and last 64 bodies compared byte for byte with variant 0's.
  python tools/ubench/variant_scan.py [circuit=compression] [n=4096] [variants...]"""
import importlib, os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
m = importlib.import_module("hot-proofs-blake3-circom_amd")
st = torch.cuda.current_stream().cuda_stream
circuit = sys.argv[1] if len(sys.argv) > 1 else "compression"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
variants = [int(v) for v in sys.argv[3:]] or ([0, 3, 8, 24, 28, 36, 52, 84, 100] if circuit == "compression" else [0, 2, 3, 24, 28, 36, 100])
os.environ.pop("B3W_VARIANT", None)
ctx0 = m.Context(circuit, 0)
recs = m.workloads.config2_compression(n) if circuit == "compression" else m.workloads.config3_nova(n)
d_recs = torch.from_numpy(recs.view(np.int32)).cuda()
body = ctx0.body_bytes
algo = n * (body + (112 if circuit == "compression" else 128))
bufs = {}
big = n * body > 16e9                                     # (SCAN_BUFFERS=placed: one buffer only; big batches: parity on a sample of bodies)
only = os.environ.get("SCAN_BUFFERS", "placed" if big else "hipMalloc,1-class,placed").split(",")
ptrs = {}
plain = None
if "hipMalloc" in only:
    plain = torch.empty(n * body, dtype=torch.uint8, device="cuda")
    ptrs["hipMalloc"] = plain.data_ptr()
if "1-class" in only:
    os.environ["B3W_PLACEMENT"] = "single"
    bufs["1-class"] = ctx0.alloc_bodies(n * body)
    os.environ.pop("B3W_PLACEMENT")
    ptrs["1-class"] = bufs["1-class"].ptr
if "placed" in only:
    bufs["placed"] = ctx0.alloc_bodies(n * body)
    ptrs["placed"] = bufs["placed"].ptr
labels = {k: v.placement for k, v in bufs.items()}
k = min(n, 64)
ref = None
print(f"{circuit} n={n}: TB/s of algorithmic bytes by variant and buffer (labels: {labels})", flush=True)
print(f"{'variant':>8s} " + " ".join(f"{b:>10s}" for b in ptrs), flush=True)
for v in variants:
    os.environ["B3W_VARIANT"] = str(v)
    ctx = m.Context(circuit, 0)
    row = []
    for name, ptr in ptrs.items():
        for _ in range(2):
            ctx.run_device(d_recs.data_ptr(), n, ptr, 0, 0, 0, st)
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ctx.run_device(d_recs.data_ptr(), n, ptr, 0, 0, 0, st)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 5)
        row.append(algo / best / 1e9)
    # byte for byte against the first variant: all three buffers, every body (public outputs and status too)
    pub = torch.zeros((n, 16), dtype=torch.int32, device="cuda"); stt = torch.full((n,), -7, dtype=torch.int32, device="cuda")
    name0, ptr0 = next(iter(ptrs.items()))
    whole = torch.empty(0)
    if plain is not None and not big:
        plain.zero_()
        ctx.run_device(d_recs.data_ptr(), n, plain.data_ptr(), 0, pub.data_ptr(), stt.data_ptr(), st)
        torch.cuda.synchronize()
        got = plain
    else:                                                  # a sample: the first and last 64 bodies and 64 from the middle, read back through a view
        ctx.run_device(d_recs.data_ptr(), n, ptr0, 0, pub.data_ptr(), stt.data_ptr(), st)
        torch.cuda.synchronize()
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        got = torch.empty(3 * k * body, dtype=torch.uint8, device="cuda")
        for j, first in enumerate((0, (n // 2) - k // 2, n - k)):
            assert hip.hipMemcpy(ctypes.c_void_p(got.data_ptr() + j * k * body), ctypes.c_void_p(ptr0 + first * body), ctypes.c_size_t(k * body), 3) == 0
    if ref is None:
        ref = (got.clone(), pub.clone(), stt.clone())
    assert torch.equal(got, ref[0]), f"variant {v}: bodies differ from variant {variants[0]}"
    assert torch.equal(pub, ref[1]) and torch.equal(stt, ref[2]), f"variant {v}: public outputs / status differ"
    print(f"{v:8d} " + " ".join(f"{r:10.3f}" for r in row), flush=True)
    ctx.close()
