"""pace_robust.py — where the fill-ordered kernel's pacing cliff stands, by batch size and kind of buffer:
B3W_FILL_PACE = sleeps + 16 x single vector-ALU steps per unit and storing wave.
  python tools/ubench/pace_robust.py [paces...]        (CIRCUIT=nova_vesta for a nova build; default compression)"""
import importlib, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
m = importlib.import_module("hot-proofs-blake3-circom_amd")
st = torch.cuda.current_stream().cuda_stream
paces = [int(a) for a in sys.argv[1:]] or [2, 16, 32, 48, 64]
circuit = os.environ.get("CIRCUIT", "compression")
nmax = 32768
recs = m.workloads.config2_compression(nmax) if circuit == "compression" else m.workloads.config3_nova(nmax)
d_recs = torch.from_numpy(recs.view(np.int32)).cuda()
os.environ["B3W_VARIANT"] = "200"
ctx = m.Context(circuit, 0)
algo = ctx.body_bytes + (112 if circuit == "compression" else 128)
bufs = {"placed": ctx.alloc_bodies(nmax * ctx.body_bytes)}
os.environ["B3W_PLACEMENT"] = "single"
bufs["one-class"] = ctx.alloc_bodies(nmax * ctx.body_bytes)
os.environ.pop("B3W_PLACEMENT")
plain = torch.empty(nmax * ctx.body_bytes, dtype=torch.uint8, device="cuda")
ptrs = {"placed (" + bufs["placed"].placement + ")": bufs["placed"].ptr, "one-class": bufs["one-class"].ptr, "torch.empty": plain.data_ptr()}
print(circuit, "TB/s by pace", paces)
for n in [int(a) for a in os.environ.get("SIZES", "1024,4096,16384,32768").split(",")]:
    for name, ptr in ptrs.items():
        row = []
        for pace in paces:
            os.environ["B3W_FILL_PACE"] = str(pace)
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
            row.append(round(n * algo / best / 1e9, 3))
        print(f"n {n:6d} {name:22s} {row}", flush=True)
