"""On-device consumer (b3w_batch_verify_device): valid witnesses check clean; any corruption of a body —
a flipped bit slot, a wrong word, a tampered input, a tampered IsZero inverse — is caught."""
import numpy as np
import pytest
import b3w_testlib as T

pytestmark = pytest.mark.gpu


def _make(m, circuit, n):
    import torch
    W = T.workloads()
    recs = W.config2_compression(n, first=300) if circuit == "compression" else W.config3_nova(n, first=300)
    ctx = m.Context(circuit, 0)
    dev = torch.device("cuda:0")
    d_recs = torch.from_numpy(recs.view(np.int32)).to(dev)
    pitch = ctx.body_bytes + 64
    d_bodies = torch.zeros((n, pitch), dtype=torch.uint8, device=dev)
    d_st = torch.zeros(n, dtype=torch.int32, device=dev)
    ctx.run_device(d_recs.data_ptr(), n, d_bodies.data_ptr(), pitch, 0, d_st.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert (d_st == 0).all().item()
    return ctx, d_bodies, pitch


def _verify(ctx, d_bodies, n, pitch):
    import torch
    d_mm = torch.full((n,), 12345, dtype=torch.int32, device=d_bodies.device)
    ctx.verify_device(d_bodies.data_ptr(), n, pitch, d_mm.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return d_mm.cpu().numpy().view(np.uint32)


@pytest.mark.parametrize("circuit", T.CIRCUITS)
def test_valid_witnesses_verify_clean_and_corruption_is_caught(circuit):
    import torch
    m = T.pkg()
    n = 37
    ctx, d_bodies, pitch = _make(m, circuit, n)
    assert (_verify(ctx, d_bodies, n, pitch) == 0).all()
    nwit = T.NWIT[circuit]
    rng = np.random.default_rng(1)
    # flip one byte in a random slot of some bodies
    victims = {3: int(rng.integers(46, nwit)), 11: nwit - 1, 20: 0, 36: int(rng.integers(46, nwit))}
    for i, slot in victims.items():
        d_bodies[i, slot * 32 + int(rng.integers(0, 32))] ^= 1
    mm = _verify(ctx, d_bodies, n, pitch)
    for i in range(n):
        assert (mm[i] >= 1) == (i in victims), (i, mm[i])
    assert all(mm[i] == 1 for i in victims)
    # tamper with an input slot (m[3] / h[1]): the recomputed witness differs nearly everywhere
    in_slot = 25 + 3 if circuit == "compression" else 18 + 1
    d_bodies[5, in_slot * 32] ^= 0x10
    mm = _verify(ctx, d_bodies, n, pitch)
    assert mm[5] > 1000 and mm[5] != 0xFFFFFFFF
    # an input slot that is not a plain 32-bit value cannot be checked here: flagged, not passed
    d_bodies[6, in_slot * 32 + 9] = 1
    assert _verify(ctx, d_bodies, n, pitch)[6] == 0xFFFFFFFF
    # padding between bodies is never read as witness data
    d_bodies[:, ctx.body_bytes:] = 0xEE
    mm2 = _verify(ctx, d_bodies, n, pitch)
    assert mm2[0] == 0 and mm2[1] == 0
    ctx.close()


def test_verify_catches_a_tampered_inverse_and_rejected_inputs():
    import torch
    m = T.pkg()
    n = 8
    ctx, d_bodies, pitch = _make(m, "nova_vesta", n)
    # w[22971] = check_root.isz.inv: a full field element for depth != 0
    d_bodies[2, 22971 * 32 + 20] ^= 0x80
    # depth (w[28]) := leaf_depth (w[15]): the circuit rejects these inputs
    d_bodies[4, 28 * 32: 28 * 32 + 4] = d_bodies[4, 15 * 32: 15 * 32 + 4]
    mm = _verify(ctx, d_bodies, n, pitch)
    assert mm[2] >= 1 and mm[4] == 0xFFFFFFFF and all(mm[i] == 0 for i in (0, 1, 3, 5, 6, 7))
    ctx.close()


def test_batch_wrapper_verify():
    m = T.pkg()
    ctx = m.Context("nova_bn254_o1", 0)
    b = m.Batch(ctx, 10)
    b.run(T.workloads().config3_nova(10, first=40))
    assert (b.verify() == 0).all()
    b.close(); ctx.close()
