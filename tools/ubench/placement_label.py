"""placement_label.py — what label does a chained-mode ring buffer get, and how fast is it really?  

  python3 tools/ubench/placement_label.py rings      the two 12.2 GB ring buffers of a nova chain: label the library reports
                                                      (B3W_PLACE_DEBUG=1 prints the allocator's own measurements on stderr) and the
                                                      nova kernel's rate into each, re-measured from outside; run it plain and under
                                                      `rocprofv3 --kernel-trace -- python3 ...` — the claim check times real launches
  python3 tools/ubench/placement_label.py plain50    50 plain hipMalloc buffers of 12.2 GB at different places of the device
                                                      (up to 16 alive at a time, spacers in between): rate of each, fraction >= 6.5 TB/s
"""
import importlib, json, os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
mode = sys.argv[1] if len(sys.argv) > 1 else "rings"
if mode == "plain50":
    os.environ["B3W_PLACEMENT"] = "plain"
else:
    os.environ["B3W_PLACE_DEBUG"] = "1"
m = importlib.import_module("hot-proofs-blake3-circom_amd")
ctx = m.Context("nova_vesta", 0)
n = 16384
recs = m.workloads.config3_nova(n)
d_recs = torch.from_numpy(recs.view(np.int32)).cuda()
d_st = torch.zeros(n, dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
nbytes = n * ctx.body_bytes


def rate(ptr, iters=5):
    for _ in range(2):
        ctx.run_device(d_recs.data_ptr(), n, ptr, 0, 0, d_st.data_ptr(), s)
    ms = ctx.time_device(d_recs.data_ptr(), n, ptr, 0, 0, d_st.data_ptr(), s, iters)
    return n * (ctx.body_bytes + 128) / ms / 1e6          # GB/s, algorithmic bytes


if mode == "rings":
    m.lib().b3w_bodies_configure(160, -1)
    out = {"under_profiler": bool(os.environ.get("ROCP_TOOL_LIBRARIES")), "buffers": []}
    bufs = [ctx.alloc_bodies(nbytes) for _ in range(2)]
    for i, b in enumerate(bufs):
        out["buffers"].append({"ring": i, "label": b.placement, "GBps": round(rate(b.ptr))})
    plain = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    out["plain_torch_empty_GBps"] = round(rate(plain.data_ptr()))
    out["stats"] = {k: int(v) for k, v in ctx.bodies_stats().items()}
    print(json.dumps(out), flush=True)
else:
    rates, alive = [], []
    spacers = []
    for i in range(50):
        if len(alive) == 16:                                # start over somewhere else: a growing spacer shifts the next run
            for b in alive:
                b.free()
            alive = []
            spacers.append(torch.empty((3 << 30) + (len(spacers) << 28), dtype=torch.uint8, device="cuda"))
        b = ctx.alloc_bodies(nbytes)
        alive.append(b)
        r = rate(b.ptr, 3)
        rates.append(round(r))
        print(f"plain buffer {i}: {b.ptr:#x} {r:.0f} GB/s", flush=True)
    fast = sum(1 for r in rates if r >= 6500)
    print(json.dumps({"plain_12GB_buffers": len(rates), "GBps": rates, "at_least_6500": fast, "fraction": fast / len(rates),
                      "min": min(rates), "max": max(rates), "median": sorted(rates)[len(rates) // 2]}), flush=True)
