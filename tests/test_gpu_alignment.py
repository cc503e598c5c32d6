"""The expand phase writes line-aligned 1 KiB tiles whose first and last lanes can lie outside the body
(csrc/b3w_kernels.hip expand()): for every body alignment (base offset x pitch mod 128) every byte of every body must
equal the oracle's and every byte outside the bodies — before the first, in the gaps of a padded pitch, after the
last — must stay untouched."""
import numpy as np
import pytest
import torch

import b3w_testlib as T

pytestmark = pytest.mark.gpu
FILL = 0xA5


@pytest.mark.parametrize("circuit,variants", [("compression", (0, 1, 2, 3, 7, 8, 22, 28, 84)), ("nova_vesta", (0, 1, 2, 3, 24, 52)), ("nova_bn254_o1", (0, 1, 23, 36))])
def test_all_alignments_bodies_exact_and_gaps_untouched(circuit, variants):
    import os
    m = T.pkg()
    n = 37                                                   # ragged for every W and every stride
    W = T.workloads()
    recs = W.config2_compression(n, first=77) if circuit == "compression" else W.config3_nova(n, first=77)
    bad, want = T.oracle_batch_u32(circuit, recs)
    assert bad == 0
    want = want.copy()
    body = T.NWIT[circuit] * 32
    dev = torch.device("cuda:0")
    d_recs = torch.from_numpy(recs.view(np.int32)).to(dev)
    margin = 4096
    for variant in variants:
        os.environ["B3W_VARIANT"] = str(variant)
        try:
            ctx = m.Context(circuit, 0)
        finally:
            del os.environ["B3W_VARIANT"]
        for pad in (0, 32, 64, 96, 128 + 32):
            pitch = body + pad
            for base_off in (0, 16, 32, 64, 96):
                buf = torch.full((2 * margin + n * pitch + 256,), FILL, dtype=torch.uint8, device=dev)
                assert buf.data_ptr() % 256 == 0
                d_st = torch.full((n,), -1, dtype=torch.int32, device=dev)
                ctx.run_device(d_recs.data_ptr(), n, buf.data_ptr() + margin + base_off, pitch, 0, d_st.data_ptr(),
                               torch.cuda.current_stream().cuda_stream)
                torch.cuda.synchronize()
                assert int(d_st.abs().sum().item()) == 0
                host = buf.cpu().numpy()
                lo = margin + base_off
                assert (host[:lo] == FILL).all(), (variant, pad, base_off, "bytes before the first body were written")
                for i in range(n):
                    got = host[lo + i * pitch: lo + i * pitch + body]
                    assert np.array_equal(got, want[i]), (variant, pad, base_off, i, np.nonzero(got != want[i])[0][:8])
                    gap = host[lo + i * pitch + body: lo + (i + 1) * pitch] if i + 1 < n else host[lo + i * pitch + body:]
                    assert (gap == FILL).all(), (variant, pad, base_off, i, "bytes after the body were written")
                if variant == variants[0]:
                    # the on-device consumer reads with the same line-aligned tiles: clean bodies verify clean, and a
                    # flipped bit in the first / last slot of a body (the masked lanes' neighbours) is caught
                    d_mm = torch.full((n,), -1, dtype=torch.int32, device=dev)
                    s = torch.cuda.current_stream().cuda_stream
                    ctx.verify_device(buf.data_ptr() + lo, n, pitch, d_mm.data_ptr(), s)
                    torch.cuda.synchronize()
                    assert int(d_mm.abs().sum().item()) == 0, (pad, base_off)
                    buf[lo + 5 * pitch] ^= 1                                  # slot 0 of body 5
                    buf[lo + 9 * pitch + body - 32] ^= 1                      # last slot of body 9
                    ctx.verify_device(buf.data_ptr() + lo, n, pitch, d_mm.data_ptr(), s)
                    torch.cuda.synchronize()
                    mm = d_mm.cpu().numpy()
                    assert mm[5] == 1 and mm[9] == 1 and mm.sum() == 2, (pad, base_off, mm[:12])
        ctx.close()
