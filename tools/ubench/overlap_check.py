"""overlap_check.py — does the constraint check of batch i overlap with the witness kernel of batch i + 1?  Two body buffers, the
witness kernel on one stream and the check on another (events between them as a ring of two would need), against the same work
on one stream."""
import importlib, os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
m = importlib.import_module("hot-proofs-blake3-circom_amd")
circuit = sys.argv[1] if len(sys.argv) > 1 else "nova_vesta"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
ctx = m.Context(circuit, 0)
r = m.R1cs(ctx)
recs = m.workloads.config2_compression(n) if circuit == "compression" else m.workloads.config3_nova(n)
d_recs = torch.from_numpy(recs.view(np.int32)).cuda()
bufs = [ctx.alloc_bodies(n * ctx.body_bytes) for _ in range(2)]
viol = [torch.zeros(n, dtype=torch.int32, device="cuda") for _ in range(2)]
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
K = 12


def serial():
    for i in range(K):
        k = i & 1
        ctx.run_device(d_recs.data_ptr(), n, bufs[k].ptr, 0, 0, 0, sa.cuda_stream)
        r.check_device(bufs[k].ptr, n, 0, viol[k].data_ptr(), 0, sa.cuda_stream)


def overlapped():
    made = [torch.cuda.Event() for _ in range(K)]
    checked = [torch.cuda.Event() for _ in range(K)]
    for i in range(K):
        k = i & 1
        if i >= 2:
            sa.wait_event(checked[i - 2])                 # the slot is free again
        ctx.run_device(d_recs.data_ptr(), n, bufs[k].ptr, 0, 0, 0, sa.cuda_stream)
        made[i].record(sa)
        sb.wait_event(made[i])
        r.check_device(bufs[k].ptr, n, 0, viol[k].data_ptr(), 0, sb.cuda_stream)
        checked[i].record(sb)


for name, fn in (("one stream", serial), ("two streams", overlapped)):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(sa); fn(); sb.wait_stream(sa); sa.wait_stream(sb); e1.record(sa); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    assert int(viol[0].abs().sum().item()) == 0 and int(viol[1].abs().sum().item()) == 0
    print(f"{circuit} n={n} {name}: {best / K:.3f} ms per batch = {n * K / best / 1e3:.2f} M steps/s (witness + check)", flush=True)
