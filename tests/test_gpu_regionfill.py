"""The fill-ordered fused kernel for caller-owned buffers (B3W_VARIANT=200, csrc/b3w_kernels.hip "REGIONFILL") against the oracle: same
bit-exact bar as the other launch shapes, plus the cases its deal has of its own — bodies that start and end anywhere inside a 128 KiB
region, regions shared by two bodies, gaps between bodies (padded pitch), buffers that start anywhere in a region, fewer bodies than
workgroup groups, many halves of the image buffer per workgroup, and the bytes around the bodies left alone."""
import os
import numpy as np
import pytest
import b3w_testlib as T

pytestmark = pytest.mark.gpu
FILL = 0x3C


@pytest.fixture(scope="module")
def m():
    return T.pkg()


def _fill_ctx(m, circuit="compression"):
    os.environ["B3W_VARIANT"] = "200"
    try:
        return m.Context(circuit, 0)
    finally:
        del os.environ["B3W_VARIANT"]


@pytest.mark.parametrize("n", [1, 2, 3, 31, 33, 97, 300])
def test_regionfill_matches_oracle_with_outputs(m, n):
    recs = T.workloads().config2_compression(n, first=123)
    bad, want = T.oracle_batch_u32("compression", recs)
    assert bad == 0
    want = want.copy()
    ctx = _fill_ctx(m)
    for pitch in (0, 771072, 770976 + 32, 770976 + 4096 + 64):
        b = m.Batch(ctx, n, pitch)
        b.run(recs)
        pub, st = b.outputs()
        assert (st == 0).all()
        assert np.array_equal(pub, want.reshape(n, -1, 32)[:, 1:17, :4].copy().view(np.uint32).reshape(n, 16)), (n, pitch)
        for i in range(n):
            got = b.fetch(i)
            assert np.array_equal(got, want[i]), (n, pitch, i, np.nonzero(got != want[i])[0][:8] // 32)
        b.close()
    ctx.close()


def test_regionfill_every_start_in_a_region_and_untouched_bytes(m):
    """The buffer starts at 32-byte steps through a 4 KiB block and at block steps through a region; pitches that put body borders
    everywhere; every byte of every body equals the oracle's, every byte outside — before the first body, in the gaps, after the last —
    stays as it was."""
    import torch
    n = 19
    recs = T.workloads().config2_compression(n, first=5)
    _, want = T.oracle_batch_u32("compression", recs)
    want = want.copy()
    ctx = _fill_ctx(m)
    body = ctx.body_bytes
    dev = torch.device("cuda:0")
    d_recs = torch.from_numpy(recs.view(np.int32)).to(dev)
    s = torch.cuda.current_stream().cuda_stream
    margin = 1 << 17
    for pad in (0, 32, 992, 4096, 131072 + 160):
        pitch = body + pad
        for skew in (0, 32, 4064, 4096, 65536 + 96, 131072 - 32):
            buf = torch.full((2 * margin + n * pitch + 4096,), FILL, dtype=torch.uint8, device=dev)
            lo = margin - (buf.data_ptr() % margin) + skew              # the first body `skew` bytes into a region
            d_st = torch.full((n,), -1, dtype=torch.int32, device=dev)
            d_pub = torch.zeros((n, 16), dtype=torch.int32, device=dev)
            ctx.run_device(d_recs.data_ptr(), n, buf.data_ptr() + lo, pitch, d_pub.data_ptr(), d_st.data_ptr(), s)
            torch.cuda.synchronize()
            assert int(d_st.abs().sum().item()) == 0
            host = buf.cpu().numpy()
            assert (host[:lo] == FILL).all(), (pad, skew, "bytes before the first body were written")
            for i in range(n):
                got = host[lo + i * pitch: lo + i * pitch + body]
                assert np.array_equal(got, want[i]), (pad, skew, i, np.nonzero(got != want[i])[0][:8])
                gap = host[lo + i * pitch + body: lo + (i + 1) * pitch] if i + 1 < n else host[lo + i * pitch + body:]
                assert (gap == FILL).all(), (pad, skew, i, "bytes after the body were written")
            assert np.array_equal(d_pub.cpu().numpy().view(np.uint32), want.reshape(n, -1, 32)[:, 1:17, :4].copy().view(np.uint32).reshape(n, 16))
    ctx.close()


def test_regionfill_refuses_what_it_cannot_take(m):
    """16-byte aligned bodies (a lane pair is one 32-byte slot) and the unsimplified nova build: B3W_E_BAD_ARGUMENT, nothing written."""
    import torch
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    ctx = _fill_ctx(m)
    recs = T.workloads().config2_compression(4)
    d_recs = torch.from_numpy(recs.view(np.int32)).to(dev)
    buf = torch.full((4 * ctx.body_bytes + 64,), FILL, dtype=torch.uint8, device=dev)
    with pytest.raises(m.B3WError):
        ctx.run_device(d_recs.data_ptr(), 4, buf.data_ptr() + 16, 0, 0, 0, s)
    torch.cuda.synchronize()
    assert bool((buf == FILL).all().item())
    ctx.close()
    nova = _fill_ctx(m, "nova_bn254_o1")                     # the unsimplified build: 11 KB images, no fill-ordered kernel
    nrec = torch.from_numpy(T.workloads().config3_nova(4).view(np.int32)).to(dev)
    nb = torch.empty(4 * nova.body_bytes, dtype=torch.uint8, device=dev)
    with pytest.raises(m.B3WError):
        nova.run_device(nrec.data_ptr(), 4, nb.data_ptr(), 0, 0, 0, s)
    nova.close()


def test_regionfill_full_config2_batch_on_a_plain_buffer_and_the_autotuner(m):
    """BASELINE config 2 (4 096 witnesses) into a caller-owned torch buffer: all public outputs against a plain BLAKE3 compression, 600
    bodies byte for byte against the oracle (the first and last of the buffer, a run across the middle), the whole buffer equal to the
    default variant's; and b3w_batch_autotune_device offers the variants on such a buffer (0, 3, 100, 200 or 201) and stays bit-exact."""
    import torch
    from test_gpu_parity import _blake3_compress_np
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    n = 4096
    recs = T.workloads().config2_compression(n)
    d_recs = torch.from_numpy(recs.view(np.int32)).to(dev)
    ctx = _fill_ctx(m)
    d_bodies = torch.full((n, ctx.body_bytes), 9, dtype=torch.uint8, device=dev)
    d_pub = torch.zeros((n, 16), dtype=torch.int32, device=dev)
    d_st = torch.full((n,), -1, dtype=torch.int32, device=dev)
    ctx.run_device(d_recs.data_ptr(), n, d_bodies.data_ptr(), 0, d_pub.data_ptr(), d_st.data_ptr(), s)
    torch.cuda.synchronize()
    assert (d_st == 0).all().item()
    want_pub = _blake3_compress_np(recs[:, 0:8], recs[:, 8:24], recs[:, 24], recs[:, 25], recs[:, 26], recs[:, 27])
    assert np.array_equal(d_pub.cpu().numpy().view(np.uint32), want_pub)
    idx = np.array(list(range(200)) + list(range(1900, 2100)) + list(range(n - 200, n)))
    _, want = T.oracle_batch_u32("compression", recs[idx])
    assert np.array_equal(d_bodies[torch.from_numpy(idx).to(dev)].cpu().numpy(), want)
    ref = m.Context("compression", 0)
    d_ref = torch.empty((n, ctx.body_bytes), dtype=torch.uint8, device=dev)
    ref.run_device(d_recs.data_ptr(), n, d_ref.data_ptr(), 0, 0, 0, s)
    torch.cuda.synchronize()
    assert torch.equal(d_ref, d_bodies)
    v, ms = ref.autotune_device(d_recs.data_ptr(), n, d_ref.data_ptr(), 0, 0, d_st.data_ptr(), s)
    assert v in (0, 3, 100, 200, 201) and ms > 0
    d_ref.fill_(1)
    ref.run_device(d_recs.data_ptr(), n, d_ref.data_ptr(), 0, 0, 0, s)
    torch.cuda.synchronize()
    assert torch.equal(d_ref, d_bodies)
    ref.close(); ctx.close()


def test_a_batch_whose_own_buffer_came_out_plain_takes_the_fill_order_and_stays_exact(m, monkeypatch):
    """b3w_batch_run under the default policy: the batch's body buffer is plain (here: B3W_PLACEMENT=plain; on a box where the placement
    search finds one class only: by itself) -> the library launches the fill-ordered kernel (the default policy asks the placement
    allocator what it knows about the buffer).  Bodies and outputs as ever; batches below 256 witnesses keep the sliced launch."""
    monkeypatch.setenv("B3W_PLACEMENT", "plain")
    n = 3000
    recs = T.workloads().config2_compression(n, first=9)
    ctx = m.Context("compression", 0)
    b = m.Batch(ctx, n)
    assert b.placement == "plain"
    b.run(recs)
    pub, st = b.outputs()
    assert (st == 0).all()
    idx = list(range(40)) + list(range(1480, 1520)) + list(range(n - 40, n))
    _, want = T.oracle_batch_u32("compression", recs[idx])
    for k, i in enumerate(idx):
        assert np.array_equal(b.fetch(i), want[k]), i
    assert np.array_equal(pub[idx], want.reshape(len(idx), -1, 32)[:, 1:17, :4].copy().view(np.uint32).reshape(len(idx), 16))
    b.run(recs[:100])
    for i in (0, 57, 99):
        assert np.array_equal(b.fetch(i), want[i] if i < 40 else T.oracle_batch_u32("compression", recs[i:i + 1])[1][0])
    b.close(); ctx.close()


@pytest.mark.parametrize("circuit", ["nova_vesta", "nova_bn254"])
def test_regionfill_nova_matches_oracle_with_rejected_steps_everywhere(m, circuit):
    """The nova O2 builds through the fill-ordered path: narrow images in the fill kernel + the 67 field inverses of every body from a
    second launch.  Rejected steps (first, middle, last; neighbours in one region) leave their bodies alone and report their status; every
    other body, the public outputs and the bytes around the bodies as ever — for several pitches and starts of the buffer in a region."""
    import torch
    n = 41
    recs = T.workloads().config3_nova(n, first=17).copy()
    bad_idx = [0, 7, 8, 23, 40]
    for i in bad_idx:
        recs[i, 14] = recs[i, 12] + (i % 2)                    # depth >= leaf_depth
    recs[11, 14] = 0xFFFFFFFF                                  # outside the kernels' domain: status 103
    nbad, want = T.oracle_batch_u32(circuit, recs)
    want = want.copy()
    ctx = _fill_ctx(m, circuit)
    body = ctx.body_bytes
    dev = torch.device("cuda:0")
    d_recs = torch.from_numpy(recs.view(np.int32)).to(dev)
    s = torch.cuda.current_stream().cuda_stream
    ref = m.Context(circuit, 0)
    d_ref_st = torch.full((n,), -1, dtype=torch.int32, device=dev)
    d_ref_pub = torch.zeros((n, 15), dtype=torch.int32, device=dev)
    d_ref = torch.full((n, body), FILL, dtype=torch.uint8, device=dev)
    ref.run_device(d_recs.data_ptr(), n, d_ref.data_ptr(), 0, d_ref_pub.data_ptr(), d_ref_st.data_ptr(), s)
    torch.cuda.synchronize()
    ref_st = d_ref_st.cpu().numpy()
    assert [i for i in range(n) if ref_st[i] != 0] == sorted(bad_idx + [11])
    margin = 1 << 17
    for pad in (0, 32, 4096 + 96):
        pitch = body + pad
        for skew in (0, 4064, 65536 + 32):
            buf = torch.full((2 * margin + n * pitch + 4096,), FILL, dtype=torch.uint8, device=dev)
            lo = margin - (buf.data_ptr() % margin) + skew
            d_st = torch.full((n,), -1, dtype=torch.int32, device=dev)
            d_pub = torch.zeros((n, 15), dtype=torch.int32, device=dev)
            ctx.run_device(d_recs.data_ptr(), n, buf.data_ptr() + lo, pitch, d_pub.data_ptr(), d_st.data_ptr(), s)
            torch.cuda.synchronize()
            st = d_st.cpu().numpy()
            assert np.array_equal(st, ref_st), (pad, skew, st)
            ok = st == 0
            assert np.array_equal(d_pub.cpu().numpy()[ok], d_ref_pub.cpu().numpy()[ok])
            host = buf.cpu().numpy()
            assert (host[:lo] == FILL).all()
            for i in range(n):
                got = host[lo + i * pitch: lo + i * pitch + body]
                if ok[i]:
                    assert np.array_equal(got, want[i]), (circuit, pad, skew, i, np.nonzero(got != want[i])[0][:8] // 32)
                else:
                    assert (got == FILL).all(), (pad, skew, i, "a rejected step's body was written")
                gap = host[lo + i * pitch + body: lo + (i + 1) * pitch] if i + 1 < n else host[lo + i * pitch + body:]
                assert (gap == FILL).all(), (pad, skew, i)
    ref.close(); ctx.close()


@pytest.mark.parametrize("circuit,n,fill_variant", [("nova_vesta", 9000, 200), ("nova_bn254", 3000, 200), ("nova_vesta", 3000, 201)])
def test_regionfill_nova_large_batch_and_large_iszero_arguments(m, circuit, n, fill_variant):
    """9 000 Vesta (3 000 BN254; 3 000 at the lighter pace) steps through the fill-ordered path against the body-stream kernel (whole buffer,
    outputs, status), among them steps whose IsZero arguments leave the table of small inverses (the second launch's general inverse) and
    steps outside the domain."""
    import torch
    recs = T.workloads().config3_nova(n, first=1).copy()
    recs[5, 12] = 250; recs[5, 13] = 4000000000; recs[5, 14] = 249          # total_depth far from depth: |k| beyond the table
    recs[77, 1] = 3000000000; recs[77, 0] = 3000000001                      # block_count near n_blocks, both huge
    rej = n // 2 + 500
    recs[rej, 14] = recs[rej, 12]                                           # rejected
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    d_recs = torch.from_numpy(recs.view(np.int32)).to(dev)
    out = {}
    for variant in (3, fill_variant):
        os.environ["B3W_VARIANT"] = str(variant)
        try:
            ctx = m.Context(circuit, 0)
        finally:
            del os.environ["B3W_VARIANT"]
        d_bodies = torch.full((n, ctx.body_bytes), FILL, dtype=torch.uint8, device=dev)
        d_pub = torch.zeros((n, 15), dtype=torch.int32, device=dev)
        d_st = torch.full((n,), -1, dtype=torch.int32, device=dev)
        ctx.run_device(d_recs.data_ptr(), n, d_bodies.data_ptr(), 0, d_pub.data_ptr(), d_st.data_ptr(), s)
        torch.cuda.synchronize()
        out[variant] = (d_bodies, d_pub, d_st)
        ctx.close()
    f = out[fill_variant]
    assert torch.equal(out[3][2], f[2]) and int(f[2][rej].item()) == 4
    ok = out[3][2] == 0
    assert torch.equal(out[3][1][ok], f[1][ok])
    assert torch.equal(out[3][0], f[0])
    idx = np.array([0, 5, 77, rej - 1, rej + 1, n - 1])
    okh = ok.cpu().numpy()
    idx = idx[okh[idx]]
    _, want = T.oracle_batch_u32(circuit, recs[idx])
    assert np.array_equal(f[0][torch.from_numpy(idx).to(dev)].cpu().numpy(), want)


def test_default_policy_by_batch_size_and_buffer(m):
    """b3w_int_default_variant through the autotuner's report for small batches and through bit-exact runs: compression on a torch buffer
    — sliced below 128 witnesses, fill-ordered from 128 on; the same batch sizes into a PLACED buffer (the allocator knows it is mixed):
    fill-ordered up to 3 072, body streams above; a 16-byte aligned buffer: never the fill order.  Every run against the oracle."""
    import torch
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    n = 3500
    recs = T.workloads().config2_compression(n, first=3)
    d_recs = torch.from_numpy(recs.view(np.int32)).to(dev)
    ctx = m.Context("compression", 0)
    body = ctx.body_bytes
    plain = torch.full((n * body + 64,), 5, dtype=torch.uint8, device=dev)
    placed = ctx.alloc_bodies(n * body)
    idx = [0, 1, 99, 255, 256, 1000, 2047, 2999, n - 1]
    _, want = T.oracle_batch_u32("compression", recs[idx])
    d_st = torch.zeros(n, dtype=torch.int32, device=dev)
    for k, expect_plain in ((100, 20 + 16), (127, 20 + 16), (128, 200), (2047, 200)):
        v, _ = ctx.autotune_device(d_recs.data_ptr(), k, plain.data_ptr(), 0, 0, d_st.data_ptr(), s)
        assert v == expect_plain, (k, v)
        v, _ = ctx.autotune_device(d_recs.data_ptr(), k, plain.data_ptr() + 16, 0, 0, d_st.data_ptr(), s)
        assert v != 200, (k, v, "16-byte aligned bodies cannot take the fill order")
    ctx2 = m.Context("compression", 0)                       # a context nobody tuned: the default policy for every size
    for ptr, name in ((plain.data_ptr(), "plain"), (placed.ptr, placed.placement), (plain.data_ptr() + 16, "unaligned")):
        for k in (100, 200, 3000, n):
            ctx2.run_device(d_recs.data_ptr(), k, ptr, 0, 0, d_st.data_ptr(), s)
            torch.cuda.synchronize()
            view = torch.empty(0)
            for j, i in enumerate(idx):
                if i < k:
                    got = np.empty(body, dtype=np.uint8)
                    import ctypes
                    hip = ctypes.CDLL("libamdhip64.so")
                    assert hip.hipMemcpy(ctypes.c_void_p(got.ctypes.data), ctypes.c_void_p(ptr + i * body), ctypes.c_size_t(body), 2) == 0
                    assert np.array_equal(got, want[j]), (name, k, i)
    placed.free(); ctx.close(); ctx2.close()
    # the nova O2 builds: the fill order from 512 steps on for a buffer the allocator does not know to be mixed, from 768 to 2 560 for one it
    # does (profiles/r06/fill_small_nova*.log)
    nv = m.Context("nova_vesta", 0)
    nrecs = torch.from_numpy(T.workloads().config3_nova(800).view(np.int32)).to(dev)
    nplain = torch.empty(800 * nv.body_bytes, dtype=torch.uint8, device=dev)
    nplaced = nv.alloc_bodies(800 * nv.body_bytes)
    for k, ptr, expect in ((511, nplain.data_ptr(), 20 + 8), (512, nplain.data_ptr(), 200), (800, nplain.data_ptr() + 16, 20 + 4)):
        v, _ = nv.autotune_device(nrecs.data_ptr(), k, ptr, 0, 0, d_st.data_ptr(), s)
        assert v == expect, (k, v)
    if nplaced.placement in ("mixed", "interleaved"):             # (both: classes alternate in it, the allocator knows)
        for k, expect in ((600, 20 + 8), (800, 200)):
            v, _ = nv.autotune_device(nrecs.data_ptr(), k, nplaced.ptr, 0, 0, d_st.data_ptr(), s)
            assert v == expect, (k, v)
    nplaced.free(); nv.close()


def test_store_only_region_shapes_stay_inside_the_buffer(m):
    """b3w_bodies_store_rate shapes 6 / 7 (the fill order with nothing but its stores, paced by sleeps / by vector-ALU instructions — what
    bench.py reads the fill-ordered kernel against on a plain buffer) write whole 128 KiB regions INSIDE [buf, buf + n * pitch) only,
    wherever the buffer starts, and refuse a buffer that holds no whole region."""
    import torch
    ctx = m.Context("compression", 0)
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    n, margin = 24, 1 << 17
    span = n * ctx.body_bytes
    for shape in (6, 7, 700, 1739):
        for skew in (0, 16, 4096 + 48, margin - 16):
            buf = torch.full((3 * margin + span,), FILL, dtype=torch.uint8, device=dev)
            lo = margin - (buf.data_ptr() % margin) + skew
            rate = ctx.store_rate(buf.data_ptr() + lo, n, 0, shape, 2, s)
            torch.cuda.synchronize()
            host = buf.cpu().numpy()
            assert rate > 10 and (host[:lo] == FILL).all() and (host[lo + span:] == FILL).all(), (shape, skew)
            first = lo + (-(buf.data_ptr() + lo)) % margin
            whole = (lo + span - first) // margin * margin
            assert (host[first:first + whole] != FILL).any(axis=None) and (host[lo:first] == FILL).all() and (host[first + whole:] == FILL).all(), (shape, skew)
            # every 16-byte store of the region walk landed: no run of 16 fill bytes left inside
            inner = host[first:first + whole].reshape(-1, 16)
            assert not (inner == FILL).all(axis=1).any(), (shape, skew)
    small = torch.zeros(4 * margin, dtype=torch.uint8, device=dev)
    with pytest.raises(m.B3WError):                                 # a pace the library was not compiled with: refused, not a launch of nothing
        ctx.store_rate(small.data_ptr(), 1, 0, 701, 1, s)
    ctx.close()


def test_regionfill_does_not_depend_on_all_workgroups_being_resident(m):
    """The kernel's 256 workgroups share nothing but the buffer: on a stream confined to 40 of the 256 CUs (hipExtStreamCreateWithCUMask) they
    run in seven rounds, no compact window, and the bodies are the same — compression and nova_vesta, against the body-stream kernel."""
    import ctypes
    import torch
    dev = torch.device("cuda:0")
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    words = (ncu + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for cu in range(40):
        mask[cu // 32] |= 1 << (cu % 32)
    hs = ctypes.c_void_p()
    assert hip.hipExtStreamCreateWithCUMask(ctypes.byref(hs), words, mask) == 0
    masked = torch.cuda.ExternalStream(hs.value)
    for circuit, n in (("compression", 700), ("nova_vesta", 700)):
        recs = T.workloads().config2_compression(n, first=9) if circuit == "compression" else T.workloads().config3_nova(n, first=9)
        d_recs = torch.from_numpy(recs.view(np.int32)).to(dev)
        ctx = _fill_ctx(m, circuit)
        a = torch.full((n, ctx.body_bytes), 1, dtype=torch.uint8, device=dev)
        b = torch.full((n, ctx.body_bytes), 2, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        ctx.run_device(d_recs.data_ptr(), n, a.data_ptr(), 0, 0, 0, masked.cuda_stream)
        os.environ["B3W_VARIANT"] = "0"
        try:
            ref0 = m.Context(circuit, 0)
        finally:
            del os.environ["B3W_VARIANT"]
        ref0.run_device(d_recs.data_ptr(), n, b.data_ptr(), 0, 0, 0, torch.cuda.current_stream().cuda_stream)
        masked.synchronize()
        torch.cuda.synchronize()
        assert torch.equal(a, b), circuit
        ctx.close(); ref0.close()
    assert hip.hipStreamDestroy(hs) == 0
