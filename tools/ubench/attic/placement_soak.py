"""placement_soak.py — many allocate / run / verify / free cycles with varying sizes and several live buffers: every body
of every buffer must verify on the device and sampled bodies must equal the oracle through hipMemcpy."""
import importlib, os, random, sys, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import b3w_testlib as T
m = importlib.import_module("hot-proofs-blake3-circom_amd")
random.seed(7)
dev = torch.device("cuda:0"); s = torch.cuda.current_stream().cuda_stream
ctxs = {c: m.Context(c, 0) for c in ("compression", "nova_vesta")}
live = []
stats = {"mixed": 0, "plain": 0}
for it in range(40):
    circuit = random.choice(list(ctxs))
    ctx = ctxs[circuit]
    n = random.choice([700, 1024, 2048, 4096, 6000, 16384])
    recs = (m.workloads.config2_compression if circuit == "compression" else m.workloads.config3_nova)(n, first=it * 100)
    d_recs = torch.from_numpy(recs.view(np.int32)).to(dev)
    d_st = torch.full((n,), -1, dtype=torch.int32, device=dev); d_mm = torch.full((n,), -1, dtype=torch.int32, device=dev)
    buf = ctx.alloc_bodies(n * ctx.body_bytes)
    stats[buf.placement] += 1
    ctx.run_device(d_recs.data_ptr(), n, buf.ptr, 0, 0, d_st.data_ptr(), s)
    ctx.verify_device(buf.ptr, n, 0, d_mm.data_ptr(), s)
    torch.cuda.synchronize()
    assert int(d_st.abs().sum().item()) == 0 and int(d_mm.abs().sum().item()) == 0, (it, circuit, n)
    idx = sorted(random.sample(range(n), 3))
    _, want = T.oracle_batch_u32(circuit, recs[idx])
    view = buf.tensor()[: n * ctx.body_bytes].view(n, ctx.body_bytes)
    for j, i in enumerate(idx):
        assert np.array_equal(view[i].cpu().numpy(), want[j]), (it, circuit, n, i)
    del view
    live.append(buf)
    while len(live) > random.choice([0, 1, 2, 3]):
        live.pop(random.randrange(len(live))).free()
    if it % 10 == 9:
        print(f"{it + 1} cycles ok, placements {stats}, free {torch.cuda.mem_get_info()[0] / 2**30:.0f} GiB", flush=True)
print("soak ok", stats)
