"""r1cs_small_slabs.py — VERDICT r02 next #3(b): would Infinity-Cache-sized slabs (256 ... 320 nova bodies, generated and checked
before they leave the 256 MB MALL) beat 65 536-body batches in the chained pass?  Times generation + check per slab size, the
slab's buffer reused (so its bytes are as cache-resident as they can be)."""
import importlib, os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
m = importlib.import_module("hot-proofs-blake3-circom_amd")
s = torch.cuda.current_stream().cuda_stream
ctx = m.Context("nova_vesta", 0)
r = m.R1cs(ctx)
total = 65536
recs = m.workloads.config3_nova(total)
d_recs = torch.from_numpy(recs.view(np.int32)).cuda()
viol = torch.zeros(total, dtype=torch.int32, device="cuda")
for n in (256, 320, 512, 1024, 4096, 16384, 65536):
    buf = torch.empty(n * ctx.body_bytes, dtype=torch.uint8, device="cuda")
    def run(check):
        for b0 in range(0, total, n):
            ctx.run_device(d_recs.data_ptr() + b0 * 128, n, buf.data_ptr(), 0, 0, 0, s)
            if check:
                r.check_device(buf.data_ptr(), n, 0, viol.data_ptr() + 4 * b0, 0, s)
    run(True)
    out = []
    for check in (False, True):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(check); e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1))
    assert int(viol.abs().sum().item()) == 0
    print(f"slab {n:6d} bodies ({n * ctx.body_bytes / 2**20:7.0f} MiB): generate {total / out[0] / 1e3:5.2f} M steps/s, generate + check {total / out[1] / 1e3:5.2f} M steps/s "
          f"(check alone {total / (out[1] - out[0]) / 1e3:5.2f} M/s)", flush=True)
    del buf
