"""The two-kernel path (TRACE -> HBM scratch -> linear SWEEP; B3W_VARIANT=100) against the oracle:
same bit-exact bar as the fused kernels, plus the cases specific to it (tiles straddling two bodies,
unaligned output base, padded pitch, several scratch chunks, rejected steps inside a tile)."""
import os
import numpy as np
import pytest
import b3w_testlib as T

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def m():
    return T.pkg()


def _sweep_ctx(m, circuit):
    os.environ["B3W_VARIANT"] = "100"
    try:
        return m.Context(circuit, 0)
    finally:
        del os.environ["B3W_VARIANT"]


@pytest.mark.parametrize("n", [1, 2, 7, 64, 300])
def test_sweep_compression_matches_oracle(m, n):
    recs = T.workloads().config2_compression(n, first=40)
    _, want = T.oracle_batch_u32("compression", recs)
    want = want.copy()
    ctx = _sweep_ctx(m, "compression")
    for pitch in (0, 771072, 770976 + 32):
        b = m.Batch(ctx, n, pitch)
        b.run(recs)
        pub, st = b.outputs()
        assert (st == 0).all()
        for i in range(n):
            got = b.fetch(i)
            assert np.array_equal(got, want[i]), (n, pitch, i, np.nonzero(got != want[i])[0][:8] // 32)
        b.close()
    ctx.close()


@pytest.mark.parametrize("circuit", ["nova_bn254", "nova_vesta", "nova_bn254_o1"])
def test_sweep_nova_matches_oracle_with_rejected_steps(m, circuit):
    recs = T.workloads().config3_nova(75, first=11).copy()
    bad_idx = [0, 9, 10, 40, 74]
    for i in bad_idx:
        recs[i, 14] = recs[i, 12] + (i % 3)            # depth >= leaf_depth
    nbad, want = T.oracle_batch_u32(circuit, recs)
    assert nbad == len(bad_idx)
    want = want.copy()
    ctx = _sweep_ctx(m, circuit)
    b = m.Batch(ctx, 75)
    b.run(recs)
    pub, st = b.outputs()
    assert [i for i in range(75) if st[i] != 0] == bad_idx
    for i in range(75):
        if i not in bad_idx:
            got = b.fetch(i)
            assert np.array_equal(got, want[i]), (circuit, i, np.nonzero(got != want[i])[0][:8] // 32)
    b.close(); ctx.close()


def test_sweep_untouched_bytes_and_unaligned_base(m):
    """Output base not 4 KiB aligned, padded pitch: padding bytes, the lead-in of the first tile and the
    bytes after the last body must be left alone; rejected steps leave their body alone."""
    import torch
    n = 21
    recs = T.workloads().config3_nova(n, first=3).copy()
    recs[4, 14] = recs[4, 12]                             # rejected
    _, want = T.oracle_batch_u32("nova_vesta", recs)
    want = want.copy()
    ctx = _sweep_ctx(m, "nova_vesta")
    body, pitch, skew = ctx.body_bytes, ctx.body_bytes + 96, 1504
    dev = torch.device("cuda:0")
    buf = torch.full((skew + n * pitch + 8192,), 0x5A, dtype=torch.uint8, device=dev)
    base = buf.data_ptr() + skew
    d_recs = torch.from_numpy(recs.view(np.int32)).to(dev)
    d_st = torch.zeros(n, dtype=torch.int32, device=dev)
    ctx.run_device(d_recs.data_ptr(), n, base, pitch, 0, d_st.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    host = buf.cpu().numpy()
    assert (host[:skew] == 0x5A).all() and (host[skew + n * pitch:] == 0x5A).all()
    for i in range(n):
        got = host[skew + i * pitch: skew + i * pitch + body]
        if i == 4:
            assert (got == 0x5A).all()
        else:
            assert np.array_equal(got, want[i]), i
        assert (host[skew + i * pitch + body: skew + (i + 1) * pitch] == 0x5A).all()
    assert d_st.cpu().numpy().tolist() == [4 if i == 4 else 0 for i in range(n)]
    ctx.close()


def test_sweep_several_scratch_chunks(m):
    """n > B3W_SWEEP_CHUNK (8192): several TRACE+SWEEP pairs; public outputs of all, sampled bodies."""
    import torch
    from test_gpu_parity import _blake3_compress_np
    n = 8192 * 2 + 777
    recs = T.workloads().config2_compression(n)
    ctx = _sweep_ctx(m, "compression")
    dev = torch.device("cuda:0")
    d_recs = torch.from_numpy(recs.view(np.int32)).to(dev)
    d_bodies = torch.empty((n, ctx.body_bytes), dtype=torch.uint8, device=dev)
    d_pub = torch.zeros((n, 16), dtype=torch.int32, device=dev)
    d_st = torch.full((n,), -1, dtype=torch.int32, device=dev)
    ctx.run_device(d_recs.data_ptr(), n, d_bodies.data_ptr(), 0, d_pub.data_ptr(), d_st.data_ptr(),
                   torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert (d_st == 0).all().item()
    want_pub = _blake3_compress_np(recs[:, 0:8], recs[:, 8:24], recs[:, 24], recs[:, 25], recs[:, 26], recs[:, 27])
    assert np.array_equal(d_pub.cpu().numpy().view(np.uint32), want_pub)
    idx = np.array([0, 1, 8190, 8191, 8192, 8193, 16383, 16384, 16385, n - 2, n - 1] + list(range(5000, 5000 + 117)))
    _, want = T.oracle_batch_u32("compression", recs[idx])
    got = d_bodies[torch.from_numpy(idx).to(dev)].cpu().numpy()
    assert np.array_equal(got, want)
    ctx.close()


def test_autotune_picks_a_variant_and_stays_bit_exact(m):
    """Large batches: the autotuner times the bit-identical candidates on the caller's buffers and keeps the fastest.  Small
    batches (up to 2 560 witnesses) follow the DEFAULT policy whatever a tuning run on a large batch chose, and the tuner then reports
    that policy's shape: the fill-ordered kernel from 128 compression witnesses on (r06: at least as fast as the sliced launch on any
    buffer), the sliced launch below (20 + waves per body)."""
    import torch
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    n = 4096
    recs = T.workloads().config2_compression(n, first=77)
    ctx = m.Context("compression", 0)
    d_recs = torch.from_numpy(recs.view(np.int32)).to(dev)
    d_bodies = torch.empty((n, ctx.body_bytes), dtype=torch.uint8, device=dev)
    d_st = torch.zeros(n, dtype=torch.int32, device=dev)
    v, ms = ctx.autotune_device(d_recs.data_ptr(), n, d_bodies.data_ptr(), 0, 0, d_st.data_ptr(), s)
    assert v in (0, 3, 100, 200, 201) and ms > 0
    _, want = T.oracle_batch_u32("compression", recs[:64])
    for k in (n, 512, 5):                                    # the tuned variant, then small batches on the same context
        d_bodies.fill_(7)
        ctx.run_device(d_recs.data_ptr(), k, d_bodies.data_ptr(), 0, 0, d_st.data_ptr(), s)
        torch.cuda.synchronize()
        assert np.array_equal(d_bodies[:min(k, 64)].cpu().numpy(), want[:min(k, 64)]), k
        if k < n:
            assert int((d_bodies[k:k + 2] != 7).sum().item()) == 0, "a small batch wrote past its bodies"
    v512, ms512 = ctx.autotune_device(d_recs.data_ptr(), 512, d_bodies.data_ptr(), 0, 0, d_st.data_ptr(), s)
    assert v512 == 200 and 0 < ms512 < 0.5                     # (fill-ordered: 0.07 ms for 512 witnesses; one body per wave took 0.14)
    v100, _ = ctx.autotune_device(d_recs.data_ptr(), 100, d_bodies.data_ptr(), 0, 0, d_st.data_ptr(), s)
    assert v100 == 20 + 16
    v1, _ = ctx.autotune_device(d_recs.data_ptr(), 1, d_bodies.data_ptr(), 0, 0, d_st.data_ptr(), s)
    assert v1 == 20 + 64
    ctx.close()
