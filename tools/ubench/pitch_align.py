"""pitch_align.py — contiguous bodies (pitch = body bytes: every body starts at another offset in a 128-byte line, its first and last line
shared with its neighbours) against a pitch rounded up to 128 bytes (whole lines throughout).  TB/s of body bytes, HIP events."""
import importlib, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
m = importlib.import_module("hot-proofs-blake3-circom_amd")
st = torch.cuda.current_stream().cuda_stream
for circuit, sizes in (("compression", (4096, 16384)), ("nova_vesta", (4096, 16384))):
    nmax = max(sizes)
    recs = m.workloads.config2_compression(nmax) if circuit == "compression" else m.workloads.config3_nova(nmax)
    d_recs = torch.from_numpy(recs.view(np.int32)).cuda()
    for variant in ("0", "200"):
        os.environ["B3W_VARIANT"] = variant
        ctx = m.Context(circuit, 0)
        os.environ.pop("B3W_VARIANT")
        body = ctx.body_bytes
        algo = body + (112 if circuit == "compression" else 128)
        aligned = (body + 127) // 128 * 128
        placed = ctx.alloc_bodies(nmax * aligned)
        plain = torch.empty(nmax * aligned, dtype=torch.uint8, device="cuda")
        for bname, ptr in (("placed " + placed.placement, placed.ptr), ("torch.empty", plain.data_ptr())):
            for n in sizes:
                row = []
                for pitch in (body, aligned):
                    for _ in range(3):
                        ctx.run_device(d_recs.data_ptr(), n, ptr, pitch, 0, 0, st)
                    best = 1e9
                    for _ in range(3):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        for _ in range(10):
                            ctx.run_device(d_recs.data_ptr(), n, ptr, pitch, 0, 0, st)
                        e1.record(); torch.cuda.synchronize()
                        best = min(best, e0.elapsed_time(e1) / 10)
                    row.append(round(n * algo / best / 1e9, 3))
                print(f"{circuit:12s} variant {variant:>3s} {bname:20s} n {n:6d}: contiguous {row[0]} | pitch {aligned} {row[1]} TB/s", flush=True)
        placed.free(); del plain; ctx.close()
