"""On-device consumer #2 (b3w_batch_commit*, csrc/b3w_commit.hip): Pedersen commitments of witness bodies against an
independent plain-integer implementation of the group law (tests/ec_ref.py), on both curves, with random and with
deliberately related generators (P + P, P - P, everything the same point), and the domain check."""
import ctypes
import time

import numpy as np
import pytest

import b3w_testlib as T
import ec_ref as E

pytestmark = pytest.mark.gpu


def _slot_values(bodies):
    """uint8 [n, nwit*32] -> list of lists of Python ints"""
    n = bodies.shape[0]
    b = bodies.reshape(n, -1, 32)
    return [[int.from_bytes(b[i, s].tobytes(), "little") for s in range(b.shape[1])] for i in range(n)]


def _key(m, ctx, curve, first_slot, gens):
    key = ctypes.c_void_p()
    buf = E.points_to_bytes(gens)
    rc = m.lib().b3w_commit_key_create(ctx.handle, E.CURVE_ID[curve], first_slot, buf, ctypes.byref(key))
    assert rc == 0, ctx.last_error()
    return key


def _commit(m, batch, key, n):
    pts = np.zeros(n * 64, dtype=np.uint8)
    st = np.zeros(n, dtype=np.int32)
    rc = m.lib().b3w_batch_commit(batch.handle, key, pts.ctypes.data, st.ctypes.data)
    assert rc == 0, batch.ctx.last_error()
    return [E.point_from_bytes(pts[64 * i: 64 * i + 64].tobytes()) for i in range(n)], st


@pytest.mark.parametrize("circuit,curve,first_slot,n", [("compression", "bn254_g1", 0, 4), ("compression", "bn254_g1", 17, 2),
                                                        ("nova_vesta", "vesta", 16, 3), ("nova_bn254", "bn254_g1", 0, 2)])
def test_commitments_match_plain_integer_group_law(circuit, curve, first_slot, n):
    m = T.pkg()
    W = T.workloads()
    recs = W.config2_compression(n, first=11) if circuit == "compression" else W.config3_nova(n, first=11)
    bad, bodies = T.oracle_batch_u32(circuit, recs)
    assert bad == 0
    vals = _slot_values(bodies.copy())
    nwit = T.NWIT[circuit]
    gens = E.random_points(curve, nwit - first_slot, seed=circuit.encode())
    assert all(E.on_curve(G, curve) for G in gens[:50])
    ctx = m.Context(circuit, 0)
    b = m.Batch(ctx, n)
    b.run(recs)
    key = _key(m, ctx, curve, first_slot, gens)
    got, st = _commit(m, b, key, n)
    assert (st == 0).all()
    for i in range(n):
        want = E.commit(vals[i][first_slot:], gens, curve)
        assert E.on_curve(got[i], curve) and got[i] == want, (circuit, i)
    m.lib().b3w_commit_key_destroy(key)
    b.close(); ctx.close()


def test_related_generators_hit_the_exceptional_cases():
    """All generators equal (every addition of a lane's second point is P + P or a multiple meeting itself), and
    generators in +/- pairs (P - P = infinity inside the sums): the result must still be the exact sum."""
    m = T.pkg()
    circuit, curve, n = "compression", "bn254_g1", 2
    p, _ = E.CURVES[curve]
    recs = T.workloads().config2_compression(n, first=3)
    _, bodies = T.oracle_batch_u32(circuit, recs)
    vals = _slot_values(bodies.copy())
    nwit = T.NWIT[circuit]
    base = E.random_points(curve, 2, seed=b"related")
    ctx = m.Context(circuit, 0)
    b = m.Batch(ctx, n)
    b.run(recs)
    # (a) one point everywhere: C = (sum of all slot values) * G
    key = _key(m, ctx, curve, 0, [base[0]] * nwit)
    got, st = _commit(m, b, key, n)
    for i in range(n):
        assert got[i] == E.mul(sum(vals[i]) % (1 << 300), base[0], p), i
    m.lib().b3w_commit_key_destroy(key)
    # (b) +G, -G, +G, -G ...: C = (sum of even slots - sum of odd slots) * G
    key = _key(m, ctx, curve, 0, [base[1] if s % 2 == 0 else E.neg(base[1], p) for s in range(nwit)])
    got, st = _commit(m, b, key, n)
    for i in range(n):
        k = sum(vals[i][0::2]) - sum(vals[i][1::2])
        want = E.mul(abs(k), base[1] if k >= 0 else E.neg(base[1], p), p)
        assert got[i] == want, i
    m.lib().b3w_commit_key_destroy(key)
    b.close(); ctx.close()


def test_commit_flags_bodies_outside_its_domain_and_reports_rate():
    import torch
    m = T.pkg()
    n = 512
    ctx = m.Context("compression", 0)
    recs = T.workloads().config2_compression(n)
    gens = E.random_points("bn254_g1", T.NWIT["compression"], seed=b"rate")
    key = _key(m, ctx, "bn254_g1", 0, gens)
    dev = torch.device("cuda:0")
    d_recs = torch.from_numpy(recs.view(np.int32)).to(dev)
    d_bodies = torch.empty((n, ctx.body_bytes), dtype=torch.uint8, device=dev)
    d_pts = torch.zeros((n, 64), dtype=torch.uint8, device=dev)
    d_st = torch.full((n,), -1, dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    ctx.run_device(d_recs.data_ptr(), n, d_bodies.data_ptr(), 0, 0, 0, s)
    L = m.lib()
    assert L.b3w_batch_commit_device(ctx.handle, key, d_bodies.data_ptr(), n, 0, d_pts.data_ptr(), d_st.data_ptr(), s) == 0
    torch.cuda.synchronize()
    assert int(d_st.abs().sum().item()) == 0
    first = d_pts.clone()
    t0 = time.perf_counter()
    for _ in range(3):
        L.b3w_batch_commit_device(ctx.handle, key, d_bodies.data_ptr(), n, 0, d_pts.data_ptr(), d_st.data_ptr(), s)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"commit: {n / dt / 1e3:.1f} k witnesses/s ({dt * 1e3:.1f} ms per {n})")
    assert torch.equal(first, d_pts)                           # deterministic
    d_bodies[7, 32 * 200] = 2                                   # slot 200 is a bit slot (outXor bits): 2 is not a bit
    L.b3w_batch_commit_device(ctx.handle, key, d_bodies.data_ptr(), n, 0, d_pts.data_ptr(), d_st.data_ptr(), s)
    torch.cuda.synchronize()
    st = d_st.cpu().numpy()
    assert st[7] == 103 and (np.delete(st, 7) == 0).all()
    L.b3w_commit_key_destroy(key)
    ctx.close()
