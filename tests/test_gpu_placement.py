"""Body-buffer placement (b3w_bodies_alloc, csrc/b3w_placement.hip): a buffer assembled from two classes of HBM
through the HIP virtual-memory API must behave exactly like a plain device buffer — same witnesses bit for bit,
copies across its 256 MiB seams, free / re-allocate — whatever placement the search ends with."""
import importlib
import os

import numpy as np
import pytest

import b3w_testlib as T

pytestmark = pytest.mark.gpu
m = importlib.import_module("hot-proofs-blake3-circom_amd")
HANDLE = 256 << 20


def seam_bodies(n, pitch):
    """indices of the bodies that straddle a 256 MiB border of the buffer, plus first and last"""
    idx = {0, n - 1}
    k = 1
    while k * HANDLE < n * pitch:
        idx.add(k * HANDLE // pitch)
        k += 1
    return sorted(i for i in idx if i < n)


def test_batch_on_placed_buffer_matches_oracle():
    n = 4096
    ctx = m.Context("compression", 0)
    recs = m.workloads.config2_compression(n)
    b = m.Batch(ctx, n)
    assert b.placement in ("mixed", "interleaved", "plain")
    b.run(recs)
    pub, st = b.outputs()
    assert (st == 0).all()
    idx = seam_bodies(n, ctx.body_bytes)
    assert len(idx) >= 10                                   # 3.16 GB = 12 pieces
    bad, want = T.oracle_batch_u32("compression", recs[idx])
    assert bad == 0
    for j, i in enumerate(idx):
        assert np.array_equal(b.fetch(i), want[j]), f"body {i} (on a seam of the placed buffer) differs from the oracle"
    assert (b.verify() == 0).all()                          # every body, on the device
    b.close()
    ctx.close()


def test_alloc_free_cycle_and_plain_override():
    ctx = m.Context("compression", 0)
    nbytes = 1024 * ctx.body_bytes                          # 790 MB: above the 512 MiB threshold
    seen = []
    for _ in range(3):
        buf = ctx.alloc_bodies(nbytes)
        assert buf.ptr and buf.ptr % (2 << 20) == 0
        seen.append(buf.placement)
        buf.free()
    small = ctx.alloc_bodies(8 * ctx.body_bytes)            # small buffers are plain hipMalloc
    assert small.placement == "plain"
    small.free()
    os.environ["B3W_PLACEMENT"] = "plain"
    try:
        buf = ctx.alloc_bodies(nbytes)
        assert buf.placement == "plain"
        buf.free()
    finally:
        del os.environ["B3W_PLACEMENT"]
    ctx.close()


def test_kernel_on_raw_placed_pointer_verifies():
    """run_device / verify_device on a b3w_bodies_alloc pointer, nova circuit (wide slots), ragged count"""
    import torch
    n = 1500
    ctx = m.Context("nova_vesta", 0)
    recs = m.workloads.config3_nova(n)
    dev = torch.device("cuda:0")
    d_recs = torch.from_numpy(recs.view(np.int32)).to(dev)
    d_pub = torch.zeros((n, ctx.public_words), dtype=torch.int32, device=dev)
    d_st = torch.full((n,), -1, dtype=torch.int32, device=dev)
    d_mm = torch.full((n,), -1, dtype=torch.int32, device=dev)
    buf = ctx.alloc_bodies(n * ctx.body_bytes)
    s = torch.cuda.current_stream().cuda_stream
    ctx.run_device(d_recs.data_ptr(), n, buf.ptr, ctx.body_bytes, d_pub.data_ptr(), d_st.data_ptr(), s)
    ctx.verify_device(buf.ptr, n, ctx.body_bytes, d_mm.data_ptr(), s)
    torch.cuda.synchronize()
    assert int(d_st.abs().sum().item()) == 0
    assert int(d_mm.abs().sum().item()) == 0
    buf.free()
    ctx.close()


def test_body_buffer_as_torch_tensor():
    """zero-copy torch view of a placed buffer (what chain.fold_witnesses hands to its consumer)"""
    import torch
    ctx = m.Context("compression", 0)
    n = 1024
    recs = m.workloads.config2_compression(n)
    buf = ctx.alloc_bodies(n * ctx.body_bytes)
    t = buf.tensor()
    assert t.data_ptr() == buf.ptr and t.numel() == buf.nbytes and t.dtype == torch.uint8
    d_recs = torch.from_numpy(recs.view(np.int32)).cuda()
    ctx.run_device(d_recs.data_ptr(), n, buf.ptr, ctx.body_bytes, 0, 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    idx = [0, 347, 348, 349, n - 1]                        # 348 = first body across the 256 MiB seam
    bad, want = T.oracle_batch_u32("compression", recs[idx])
    view = t[: n * ctx.body_bytes].view(n, ctx.body_bytes)
    for j, i in enumerate(idx):
        assert np.array_equal(view[i].cpu().numpy(), want[j])
    del t, view
    buf.free()
    ctx.close()


def test_recycled_buffers_hold_what_the_kernel_wrote():
    """Allocate / run / read back through hipMemcpy / free, five times over: the pieces of a freed buffer are reused
    by the next one.  (On this ROCm stack a virtual address that is unmapped and mapped to another physical handle
    keeps serving the old pages — tools/ubench/vmm_remap.hip — which is why b3w_placement.hip never maps an address
    twice; an allocator that did would fail exactly here: the kernel's stores and the copy engine's reads would see
    different memory.)"""
    ctx = m.Context("compression", 0)
    n = 1024
    idx = [0, 347, 348, 349, n - 1]
    for rnd in range(5):
        recs = m.workloads.config2_compression(n, first=5000 * rnd)
        b = m.Batch(ctx, n)
        b.run(recs)
        bad, want = T.oracle_batch_u32("compression", recs[idx])
        assert bad == 0
        for j, i in enumerate(idx):
            assert np.array_equal(b.fetch(i), want[j]), (rnd, i)
        assert (b.verify() == 0).all()
        b.close()
    ctx.close()


def test_wtns_writer_streams_across_a_seam_of_a_placed_buffer(tmp_path):
    """b3w_batch_write_wtns (pinned double-buffered hipMemcpy2DAsync) out of a placed buffer, bodies 340..359 straddle the
    first 256 MiB seam (body 348): files byte-identical to calculateWTNSBin's image."""
    ctx = m.Context("compression", 0)
    n = 1024
    recs = m.workloads.config2_compression(n, first=77)
    b = m.Batch(ctx, n)
    b.run(recs)
    wrote = b.write_wtns(tmp_path, "w_", first=340, count=20)
    assert wrote == 20
    idx = [340, 347, 348, 349, 359]
    bad, want = T.oracle_batch_u32("compression", recs[idx])
    hdr = ctx.wtns_header()
    for j, i in enumerate(idx):
        assert (tmp_path / f"w_{i}.wtns").read_bytes() == hdr + want[j].tobytes(), i
    b.close()
    ctx.close()


def _fill_rate(ctx, ptr, n, d_recs, d_st, torch):
    """GB/s of the witness kernel writing n bodies at ptr (2 warm-ups, 6 timed launches)"""
    s = torch.cuda.current_stream().cuda_stream
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(2):
        ctx.run_device(d_recs.data_ptr(), n, ptr, 0, 0, d_st.data_ptr(), s)
    e0.record()
    for _ in range(6):
        ctx.run_device(d_recs.data_ptr(), n, ptr, 0, 0, d_st.data_ptr(), s)
    e1.record()
    torch.cuda.synchronize()
    return 6 * n * ctx.body_bytes / 1e6 / e0.elapsed_time(e1)


def test_mixed_is_a_measured_claim():
    """A buffer reported "mixed" has been timed with the real witness kernel against plain memory inside b3w_bodies_alloc
    (demoted below +10 %); re-measured here from outside: every mixed buffer beats a plain hipMalloc buffer by > 10 %."""
    import torch
    ctx = m.Context("compression", 0)
    n = 4096
    d_recs = torch.from_numpy(m.workloads.config2_compression(n).view(np.int32)).cuda()
    d_st = torch.zeros(n, dtype=torch.int32, device="cuda")
    # (the yardstick the allocator itself uses: the slowest of three distinct plain buffers — one hipMalloc buffer in eight or so
    # straddles a class border by luck and is as fast as a placed one; r05: this test met one, 6 910 GB/s)
    keep, r_plain = [], None
    for _ in range(3):
        plain = torch.empty(n * ctx.body_bytes, dtype=torch.uint8, device="cuda")
        r = _fill_rate(ctx, plain.data_ptr(), n, d_recs, d_st, torch)
        r_plain = r if r_plain is None else min(r_plain, r)
        keep.append(plain)
    buf = ctx.alloc_bodies(n * ctx.body_bytes)
    r_buf = _fill_rate(ctx, buf.ptr, n, d_recs, d_st, torch)
    print(f"plain {r_plain:.0f} GB/s, b3w_bodies_alloc ({buf.placement}) {r_buf:.0f} GB/s")
    if buf.placement == "mixed":
        assert r_buf > 1.08 * r_plain                      # the allocator's own threshold is 1.10; timing noise allowed for
    st = ctx.bodies_stats()
    assert st["live_buffers"] >= 1 and st["live_bytes"] >= n * ctx.body_bytes and st["arena_used"] <= st["arena_bytes"]
    buf.free()
    del plain, keep
    ctx.close()


def test_service_2000_buffers_of_mixed_sizes_keep_their_rate_and_the_allocator_stays_bounded():
    """A long-lived process (r04 verdict #8): 2 000 allocate / fill / free cycles of body buffers of mixed sizes — 0.6 to 6.3 GB, now and
    then two alive at once — on one context.  The last buffers are as fast as the first (within 3 %), every buffer checked holds
    what the kernel wrote, and b3w_bodies_stats stays bounded all the way: pooled memory under its cap, no live buffer left behind,
    physical handles created in the hundreds (the pool is reused, not re-walked), address space used up = the sum of the buffers'
    sizes (never reused: about a fifth of the 32 TiB arena after 2 000 buffers)."""
    import torch
    ctx = m.Context("compression", 0)
    n = 4096
    recs = m.workloads.config2_compression(2 * n)
    d_recs = torch.from_numpy(recs.view(np.int32)).cuda()
    d_st = torch.zeros(2 * n, dtype=torch.int32, device="cuda")
    d_mm = torch.zeros(2 * n, dtype=torch.int32, device="cuda")
    rng = np.random.default_rng(8)
    sizes = rng.choice([768, 1024, 2048, 3000, 4096, 6000, 8192], size=2000)     # bodies per buffer
    sizes[:3] = n
    sizes[-3:] = n
    rates, labels, total = [], [], 0
    st0 = ctx.bodies_stats()
    held = None
    for k, nb in enumerate(int(x) for x in sizes):
        buf = ctx.alloc_bodies(nb * ctx.body_bytes)
        total += -(-nb * ctx.body_bytes // HANDLE) * HANDLE
        labels.append(buf.placement)
        if k < 3 or k >= 1997:
            rates.append(_fill_rate(ctx, buf.ptr, nb, d_recs, d_st, torch))
        elif k % 10 == 0:
            ctx.run_device(d_recs.data_ptr(), nb, buf.ptr, 0, 0, d_st.data_ptr(), torch.cuda.current_stream().cuda_stream)
        if k % 100 == 0 or k == 1999:
            ctx.run_device(d_recs.data_ptr(), nb, buf.ptr, 0, 0, d_st.data_ptr(), torch.cuda.current_stream().cuda_stream)
            ctx.verify_device(buf.ptr, nb, 0, d_mm.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            assert int(d_mm[:nb].abs().sum().item()) == 0 and int(d_st[:nb].abs().sum().item()) == 0, k
        st = ctx.bodies_stats()
        assert st["pooled_bytes"] <= 13 << 30, (k, st)
        assert st["live_buffers"] == (2 if held is not None else 1), (k, st)
        if held is not None:
            held.free()
            held = None
        if k % 7 == 3 and k < 1990:
            held = buf                                      # stays alive over the next allocation
        else:
            buf.free()
    st = ctx.bodies_stats()
    print(f"rates GB/s first {[round(r) for r in rates[:3]]} last {[round(r) for r in rates[3:]]}; labels {sorted(set(labels))}; arena used "
          f"{(st['arena_used'] - st0['arena_used']) / 2**40:.2f} TiB of {st['arena_bytes'] / 2**40:.0f}; handles created {st['handles_created']}; pooled {st['pooled_bytes'] >> 30} GiB")
    assert st["live_buffers"] == 0 and st["live_bytes"] == 0
    # placed all the way, or plain all the way (a box without a second class): never a placed buffer early and plain ones later.
    # ("interleaved" = placed, but that size's real-kernel check found this box's plain buffer within 10 %: the small buffers')
    assert set(labels) <= {"mixed", "interleaved"} or set(labels) == {"plain"}, f"the placement changed along the way: {sorted(set(labels))}"
    first, last = max(rates[:3]), max(rates[3:])
    assert abs(last - first) <= 0.03 * first, (first, last)
    assert st["arena_used"] - st0["arena_used"] <= total + (256 << 30)         # the buffers themselves + the probe mappings of the handles ever created
    assert st["handles_created"] - st0["handles_created"] <= 8192               # new physical memory per cycle stays a fraction of a buffer: the pool serves the rest
    ctx.close()


def test_address_space_used_up_means_plain_buffers_not_failures():
    """The placement arena only grows (virtual addresses are never reused on this ROCm stack).  When it is used up — here a 12 GiB
    arena instead of 32 TiB: 48 slots of 256 MiB, a 1.6 GB buffer takes seven and a search some more — b3w_bodies_alloc keeps answering: the buffer comes from hipMalloc, is labelled plain, and holds
    what the kernel writes; a batch never fails for it."""
    import subprocess, sys
    script = r"""
import importlib, sys
import numpy as np, torch
sys.path.insert(0, %r)
m = importlib.import_module("hot-proofs-blake3-circom_amd")
ctx = m.Context("compression", 0)
n = 2048
d_recs = torch.from_numpy(m.workloads.config2_compression(n).view(np.int32)).cuda()
d_st = torch.zeros(n, dtype=torch.int32, device="cuda"); d_mm = torch.zeros(n, dtype=torch.int32, device="cuda")
labels = []
for k in range(12):
    buf = ctx.alloc_bodies(n * ctx.body_bytes)
    labels.append(buf.placement)
    ctx.run_device(d_recs.data_ptr(), n, buf.ptr, 0, 0, d_st.data_ptr(), 0)
    ctx.verify_device(buf.ptr, n, 0, d_mm.data_ptr(), 0)
    torch.cuda.synchronize()
    assert int(d_mm.abs().sum().item()) == 0 and int(d_st.abs().sum().item()) == 0, k
    buf.free()
b = m.Batch(ctx, 1024)                                   # a batch object's own buffer, after the arena has run out
b.run(m.workloads.config2_compression(1024))
st = ctx.bodies_stats()
print("RESULT", ",".join(labels), b.placement, st["arena_bytes"] >> 30, st["arena_used"] >> 30)
""" % T.ROOT
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=600, cwd=T.ROOT,
                       env=dict(os.environ, B3W_PLACE_ARENA_GIB="12"))
    assert r.returncode == 0, r.stderr[-2000:]
    line = [x for x in r.stdout.splitlines() if x.startswith("RESULT")][0].split()
    labels = line[1].split(",")
    assert int(line[3]) == 12 and int(line[4]) <= 12
    assert labels[0] in ("mixed", "interleaved", "plain") and labels[-1] == "plain" and "plain" in labels[3:], labels
    assert line[2] in ("mixed", "interleaved", "plain"), line      # (the batch's own, smaller buffer: placed while the arena still has room for it)


def test_chain_rings_are_reused_across_chain_objects():
    """b3w_chain_create after b3w_chain_destroy on one context takes the destroyed chain's ring buffers: no new address
    space, same pointers."""
    import ctypes
    ctx = m.Context("nova_vesta", 0)
    L = m.lib()
    data = (np.arange(64 * 1024) % 251).astype(np.uint8)

    def chain(nbytes):
        h = ctypes.c_void_p()
        assert L.b3w_chain_create(ctx.handle, nbytes, 0, L.b3w_chain_num_chunks(nbytes), 1024, 2, 1, ctypes.byref(h)) == 0
        return h
    h = chain(64 * 1024)
    assert L.b3w_chain_run_leaves(h, data.ctypes.data, None, None, None) == 0 and L.b3w_chain_run_parents(h, None, None, None, None) == 0
    used = ctx.bodies_stats()["arena_used"]
    L.b3w_chain_destroy(h)
    for nbytes in (32 * 1024, 64 * 1024, 5 * 1024):       # other preimages, same ring geometry
        h = chain(nbytes)
        assert L.b3w_chain_run_leaves(h, data.ctypes.data, None, None, None) == 0 and L.b3w_chain_run_parents(h, None, None, None, None) == 0
        root = np.zeros(8, dtype=np.uint32)
        assert L.b3w_chain_outputs(h, None, None, root.ctypes.data, None) == 0
        import blake3_ref as B
        assert list(root) == B.hash_words(data[:nbytes].tobytes())
        L.b3w_chain_destroy(h)
        assert ctx.bodies_stats()["arena_used"] == used
    ctx.close()


def _concurrent_searcher(rank, ret):
    import time
    import torch
    torch.cuda.set_device(0)
    mm = importlib.import_module("hot-proofs-blake3-circom_amd")
    mm.lib().b3w_bodies_search_limit(20.0)
    ctx = mm.Context("compression", 0)
    n = 2048                                                # 1.58 GB per process
    t0 = time.perf_counter()
    buf = ctx.alloc_bodies(n * ctx.body_bytes)
    alloc_s = time.perf_counter() - t0
    recs = mm.workloads.config2_compression(n, first=1000 * rank)
    dev = torch.device("cuda:0")
    d_recs = torch.from_numpy(recs.view(np.int32)).to(dev)
    d_st = torch.full((n,), -1, dtype=torch.int32, device=dev)
    d_mm = torch.full((n,), -1, dtype=torch.int32, device=dev)
    ctx.run_device(d_recs.data_ptr(), n, buf.ptr, 0, 0, d_st.data_ptr(), 0)
    ctx.verify_device(buf.ptr, n, ctx.body_bytes, d_mm.data_ptr(), 0)
    torch.cuda.synchronize()
    ret[rank] = dict(placement=buf.placement, alloc_s=alloc_s, ok=int(d_st.abs().sum().item()) == 0 and int(d_mm.abs().sum().item()) == 0,
                     cost=ctx.placement_cost())
    buf.free()
    ctx.close()


def test_two_processes_search_the_same_gpu_at_once():
    """round-3 verdict #6: what a shared box does — two processes run the placement search on ONE GPU side by side (each other's
    probes disturb the timings, each other's handles break up the linear hand-out).  Both must come back inside the search's time
    limit with a usable buffer that says honestly what it is, and with what the search cost."""
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_concurrent_searcher, args=(ret,), nprocs=2, join=True)
    for r in (0, 1):
        got = ret[r]
        assert got["ok"] and got["placement"] in ("mixed", "interleaved", "plain"), got
        c = got["cost"]
        assert c["search_limit_s"] == 20.0 and 0 <= c["search_s"] <= got["alloc_s"] + 0.5
        assert got["alloc_s"] < 20.0 + 40.0, got            # the limit, plus seam checks and the real-kernel check of a "mixed" claim
        assert c["search_gib_walked"] >= 1.5                # at least the buffer itself was created by the search
        assert c["search_timeouts"] in (0, 1)


def test_search_time_limit_ends_in_a_plain_buffer_that_says_so():
    """a search that may take no time at all: the buffer is plain, the time-out is counted, nothing fails"""
    import subprocess, sys
    script = r"""
import importlib, sys
sys.path.insert(0, %r)
m = importlib.import_module("hot-proofs-blake3-circom_amd")
m.lib().b3w_bodies_search_limit(1e-6)
ctx = m.Context("compression", 0)
buf = ctx.alloc_bodies(1024 * ctx.body_bytes)
print("RESULT", buf.placement, ctx.placement_cost()["search_timeouts"])
""" % T.ROOT
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=300, cwd=T.ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [x for x in r.stdout.splitlines() if x.startswith("RESULT")][0].split()
    assert line[1] == "plain" and int(line[2]) == 1
