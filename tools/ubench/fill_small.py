"""fill_small.py — small batches: the default policy's launch against the fill-ordered kernel forced (B3W_VARIANT=200), on a placed and on a
one-class buffer; FRESH windows of a large buffer per launch (a repeated small batch lives in the Infinity Cache).
  python tools/ubench/fill_small.py [circuit=compression]"""
import importlib, os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
m = importlib.import_module("hot-proofs-blake3-circom_amd")
st = torch.cuda.current_stream().cuda_stream
circuit = sys.argv[1] if len(sys.argv) > 1 else "compression"
nbuf = 16384
recs = m.workloads.config2_compression(nbuf) if circuit == "compression" else m.workloads.config3_nova(nbuf)
d_recs = torch.from_numpy(recs.view(np.int32)).cuda()
ctxs = {}
os.environ.pop("B3W_VARIANT", None)
ctxs["default"] = m.Context(circuit, 0)
os.environ["B3W_VARIANT"] = "200"
ctxs["fill"] = m.Context(circuit, 0)
os.environ.pop("B3W_VARIANT")
bufs = {"placed": ctxs["default"].alloc_bodies(nbuf * ctxs["default"].body_bytes)}
os.environ["B3W_PLACEMENT"] = "single"
bufs["1-class"] = ctxs["default"].alloc_bodies(nbuf * ctxs["default"].body_bytes)
os.environ.pop("B3W_PLACEMENT")
body = ctxs["default"].body_bytes
algo = body + (112 if circuit == "compression" else 128)
print(circuit, "n: TB/s  default placed | fill placed | default 1-class | fill 1-class")
for n in (64, 128, 256, 512, 768, 1024, 1536, 2048, 2560, 3072, 4096):
    row = []
    for bname in ("placed", "1-class"):
        for cname in ("default", "fill"):
            ctx, buf = ctxs[cname], bufs[bname]
            wins = nbuf // n
            for k in range(2):
                ctx.run_device(d_recs.data_ptr(), n, buf.ptr + (k % wins) * n * body, 0, 0, 0, st)
            reps = max(8, min(64, wins))
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for k in range(reps):
                    ctx.run_device(d_recs.data_ptr(), n, buf.ptr + (k % wins) * n * body, 0, 0, 0, st)
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / reps)
            row.append(n * algo / best / 1e9)
    print(f"{n:5d}: {row[0]:7.3f} | {row[1]:7.3f} | {row[2]:7.3f} | {row[3]:7.3f}", flush=True)
