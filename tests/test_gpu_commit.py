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


def _key(m, ctx, curve, first_slot, gens, window=0):
    """window: 12, 16 or 18 bits per table window; 0 = the library's choice (b3w_commit_key_create)"""
    key = ctypes.c_void_p()
    buf = E.points_to_bytes(gens)
    if window:
        rc = m.lib().b3w_commit_key_create_ex(ctx.handle, E.CURVE_ID[curve], first_slot, buf, window, ctypes.byref(key))
    else:
        rc = m.lib().b3w_commit_key_create(ctx.handle, E.CURVE_ID[curve], first_slot, buf, ctypes.byref(key))
    assert rc == 0, ctx.last_error()
    assert m.lib().b3w_commit_key_window(key) == (window or m.lib().b3w_commit_key_window(key)) and \
        m.lib().b3w_commit_key_window(key) in (12, 16, 18)
    return key


def _commit(m, batch, key, n):
    pts = np.zeros(n * 64, dtype=np.uint8)
    st = np.zeros(n, dtype=np.int32)
    rc = m.lib().b3w_batch_commit(batch.handle, key, pts.ctypes.data, st.ctypes.data)
    assert rc == 0, batch.ctx.last_error()
    return [E.point_from_bytes(pts[64 * i: 64 * i + 64].tobytes()) for i in range(n)], st


@pytest.mark.parametrize("circuit,curve,first_slot,n,window", [
    ("compression", "bn254_g1", 0, 4, 12), ("compression", "bn254_g1", 17, 2, 16), ("nova_vesta", "vesta", 16, 3, 12),
    ("nova_vesta", "vesta", 0, 2, 16), ("nova_bn254", "bn254_g1", 0, 2, 0), ("nova_vesta", "vesta", 0, 2, 18), ("compression", "bn254_g1", 5, 2, 18)])
def test_commitments_match_plain_integer_group_law(circuit, curve, first_slot, n, window):
    m = T.pkg()
    W = T.workloads()
    recs = W.config2_compression(n, first=11) if circuit == "compression" else W.config3_nova(n, first=11)
    bad, bodies = T.oracle_batch_u32(circuit, recs)
    assert bad == 0
    vals = _slot_values(bodies.copy())
    nwit = T.NWIT[circuit]
    gens = E.random_points(curve, nwit - first_slot)
    assert all(E.on_curve(G, curve) for G in gens[:50])
    ctx = m.Context(circuit, 0)
    b = m.Batch(ctx, n)
    b.run(recs)
    key = _key(m, ctx, curve, first_slot, gens, window)
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
    for window in (12, 16, 18):
        # (a) one point everywhere: C = (sum of all slot values) * G
        key = _key(m, ctx, curve, 0, [base[0]] * nwit, window)
        got, st = _commit(m, b, key, n)
        for i in range(n):
            assert got[i] == E.mul(sum(vals[i]) % (1 << 300), base[0], p), (window, i)
        m.lib().b3w_commit_key_destroy(key)
        # (b) +G, -G, +G, -G ...: C = (sum of even slots - sum of odd slots) * G
        key = _key(m, ctx, curve, 0, [base[1] if s % 2 == 0 else E.neg(base[1], p) for s in range(nwit)], window)
        got, st = _commit(m, b, key, n)
        for i in range(n):
            k = sum(vals[i][0::2]) - sum(vals[i][1::2])
            want = E.mul(abs(k), base[1] if k >= 0 else E.neg(base[1], p), p)
            assert got[i] == want, (window, i)
        m.lib().b3w_commit_key_destroy(key)
    bad = ctypes.c_void_p()
    assert m.lib().b3w_commit_key_create_ex(ctx.handle, 0, 0, E.points_to_bytes([base[0]] * nwit), 13, ctypes.byref(bad)) == 100
    b.close(); ctx.close()


def test_commit_flags_bodies_outside_its_domain_and_reports_rate():
    import torch
    m = T.pkg()
    n = 512
    ctx = m.Context("compression", 0)
    recs = T.workloads().config2_compression(n)
    gens = E.random_points("bn254_g1", T.NWIT["compression"])
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
    d_bodies[9, 32 * 25 + 9] = 1                                # slot 25 = m[0], a 32-bit slot: a byte beyond its width
    L.b3w_batch_commit_device(ctx.handle, key, d_bodies.data_ptr(), n, 0, d_pts.data_ptr(), d_st.data_ptr(), s)
    torch.cuda.synchronize()
    st = d_st.cpu().numpy()
    assert st[7] == 103 and st[9] == 103 and (np.delete(st, [7, 9]) == 0).all()
    L.b3w_commit_key_destroy(key)
    ctx.close()


def test_chained_pass_with_the_commit_consumer():
    """preimage -> every step witness (native chained pass) -> one commitment per step, nothing but 64-byte points kept:
    b3w_commit_consumer as the ring's consumer.  Checked for sampled steps against the plain-integer group law on the
    oracle's witness of the same step record, and the pass still yields BLAKE3(preimage)."""
    import torch, blake3_ref

    class Sink(ctypes.Structure):
        _fields_ = [("ctx", ctypes.c_void_p), ("key", ctypes.c_void_p), ("d_points", ctypes.c_void_p), ("d_status", ctypes.c_void_p),
                    ("error", ctypes.c_int32)]
    m = T.pkg()
    L = m.lib()
    circuit, curve = "nova_vesta", "vesta"
    ctx = m.Context(circuit, 0)
    data = ((np.arange(8 * 1024, dtype=np.uint64) * 40503 + 11) % 253).astype(np.uint8)        # 8 chunks: 128 leaf + 24 parent steps
    gens = E.random_points(curve, T.NWIT[circuit])
    key = _key(m, ctx, curve, 0, gens)
    h = ctypes.c_void_p()
    assert L.b3w_chain_create(ctx.handle, data.size, 0, 8, 48, 2, 1, ctypes.byref(h)) == 0      # batches of 48 steps: several ring turns
    nl, npar = ctypes.c_uint64(), ctypes.c_uint64()
    L.b3w_chain_info(h, ctypes.byref(nl), ctypes.byref(npar), None, None, None)
    steps = nl.value + npar.value
    assert (nl.value, npar.value) == (128, 24)
    dev = torch.device("cuda:0")
    d_pts = torch.zeros((steps, 64), dtype=torch.uint8, device=dev)
    d_st = torch.full((steps,), -1, dtype=torch.int32, device=dev)
    sink = Sink(ctx.handle, key, d_pts.data_ptr(), d_st.data_ptr(), 0)
    consumer = ctypes.cast(L.b3w_commit_consumer, ctypes.c_void_p)
    assert L.b3w_chain_run_leaves(h, data.ctypes.data, consumer, ctypes.byref(sink), None) == 0
    assert L.b3w_chain_run_parents(h, None, consumer, ctypes.byref(sink), None) == 0
    recs = np.zeros((steps, 32), dtype=np.uint32)
    root = np.zeros(8, dtype=np.uint32)
    assert L.b3w_chain_outputs(h, None, None, root.ctypes.data, None) == 0
    torch.cuda.synchronize()
    assert sink.error == 0 and int(d_st.abs().sum().item()) == 0
    assert root.tobytes() == blake3_ref.blake3(data.tobytes())
    import ctypes as C
    hip_recs = m.chain._view(L.b3w_chain_records(h), (steps, 32), "<i4", dev).cpu().numpy().view(np.uint32)
    idx = [0, 47, 48, 127, 128, 151]                       # batch borders, first parent step, last step
    bad, bodies = T.oracle_batch_u32(circuit, hip_recs[idx])
    assert bad == 0
    vals = _slot_values(bodies.copy())
    pts = d_pts.cpu().numpy()
    for j, i in enumerate(idx):
        assert E.point_from_bytes(pts[i].tobytes()) == E.commit(vals[j], gens, curve), i
    L.b3w_chain_destroy(h)
    L.b3w_commit_key_destroy(key)
    ctx.close()


def test_chained_pass_check_then_commit():
    """witness -> constraint check -> commitment for every step of a preimage, natively: b3w_r1cs_consumer (the derived
    system of the Vesta O2 build) chained to b3w_commit_consumer as the ring's consumer.  Every step satisfies the step
    circuit, the points equal those of the commit consumer alone, the fold ends in BLAKE3(preimage)."""
    import torch, blake3_ref
    CONSUMER = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_void_p)

    class CommitSink(ctypes.Structure):
        _fields_ = [("ctx", ctypes.c_void_p), ("key", ctypes.c_void_p), ("d_points", ctypes.c_void_p), ("d_status", ctypes.c_void_p),
                    ("error", ctypes.c_int32)]

    class R1csSink(ctypes.Structure):
        _fields_ = [("ctx", ctypes.c_void_p), ("r1cs", ctypes.c_void_p), ("d_violations", ctypes.c_void_p), ("next", ctypes.c_void_p),
                    ("next_user", ctypes.c_void_p), ("error", ctypes.c_int32)]
    m = T.pkg()
    L = m.lib()
    circuit, curve = "nova_vesta", "pallas"
    ctx = m.Context(circuit, 0)
    r1cs = m.R1cs(ctx)
    data = ((np.arange(5 * 1024 + 300, dtype=np.uint64) * 9973 + 5) % 251).astype(np.uint8)   # 6 chunks, incomplete tree: 85 leaf + 16 parent steps
    key = m.CommitKey(ctx, curve, E.points_to_bytes(E.random_points(curve, T.NWIT[circuit])), window=12)
    dev = torch.device("cuda:0")

    def run(with_check):
        h = ctypes.c_void_p()
        assert L.b3w_chain_create(ctx.handle, data.size, 0, 6, 32, 2, 1, ctypes.byref(h)) == 0
        nl, npar = ctypes.c_uint64(), ctypes.c_uint64()
        L.b3w_chain_info(h, ctypes.byref(nl), ctypes.byref(npar), None, None, None)
        steps = nl.value + npar.value
        d_pts = torch.zeros((steps, 64), dtype=torch.uint8, device=dev)
        d_st = torch.full((steps,), -1, dtype=torch.int32, device=dev)
        d_viol = torch.full((steps,), -1, dtype=torch.int32, device=dev)
        csink = CommitSink(ctx.handle, key.handle, d_pts.data_ptr(), d_st.data_ptr(), 0)
        commit_fn = ctypes.cast(L.b3w_commit_consumer, ctypes.c_void_p)
        if with_check:
            rsink = R1csSink(ctx.handle, r1cs.handle, d_viol.data_ptr(), commit_fn, ctypes.cast(ctypes.byref(csink), ctypes.c_void_p), 0)
            fn, user = ctypes.cast(L.b3w_r1cs_consumer, ctypes.c_void_p), ctypes.byref(rsink)
        else:
            rsink, fn, user = None, commit_fn, ctypes.byref(csink)
        assert L.b3w_chain_run_leaves(h, data.ctypes.data, fn, user, None) == 0
        assert L.b3w_chain_run_parents(h, None, fn, user, None) == 0
        root = np.zeros(8, dtype=np.uint32)
        assert L.b3w_chain_outputs(h, None, None, root.ctypes.data, None) == 0
        torch.cuda.synchronize()
        assert csink.error == 0 and (rsink is None or rsink.error == 0)
        out = (steps, nl.value, npar.value, root.tobytes(), d_pts.cpu().numpy().copy(), d_st.cpu().numpy().copy(), d_viol.cpu().numpy().copy())
        L.b3w_chain_destroy(h)
        return out
    steps, nl, npar, root, pts, st, viol = run(True)
    assert (nl, npar) == (5 * 16 + 5, 4 * 3 + 2 * 2) and root == blake3_ref.blake3(data.tobytes())
    assert (viol == 0).all() and (st == 0).all()                    # every step witness satisfies all 23 744 constraints
    _, _, _, root2, pts2, st2, viol2 = run(False)
    assert root2 == root and np.array_equal(pts, pts2) and (viol2 == -1).all()
    key.close(); r1cs.close(); ctx.close()


@pytest.mark.parametrize("circuit,curve,window", [("compression", "bn254_g1", 16), ("nova_vesta", "vesta", 12), ("nova_bn254", "bn254_g1", 16),
                                                  ("nova_bn254_o1", "bn254_g1", 12)])
def test_commitments_from_records_equal_commitments_of_the_bodies(circuit, curve, window):
    """b3w_commit_records_device (bits taken from the TRACE images, no bodies) against b3w_batch_commit_device on the
    bodies of the same records — including rejected nova steps — and against the plain-integer group law."""
    import torch
    m = T.pkg()
    W = T.workloads()
    n = 300                                                   # ragged against every tile of the kernels
    recs = W.config2_compression(n, first=5) if circuit == "compression" else W.config3_nova(n, first=5)
    if circuit != "compression":
        recs = recs.copy()
        recs[7, 14] = recs[7, 12]                             # depth = leaf_depth: CheckDepth rejects the step
        # IsZero arguments beyond the 2 047-entry table of tabulated inverse points (r03: the O2 builds add ONE point per gadget in
        # records mode): total_depth 40 000 puts k = total_depth - i - 2 - depth past it for every eqs[i] — those slots must then go
        # through the bit windows like any other; n_blocks - 1 - block_count likewise
        recs[11, 13] = 40000
        recs[12, 13] = 2049 + recs[12, 14] + 2                 # k = 2 049, 2 048, 2 047, ...: both sides of the table's edge
        recs[13, 0] = 5000; recs[13, 1] = 3                   # n_blocks - 1 - block_count = 4 996
    ctx = m.Context(circuit, 0)
    first_slot = 3
    gens = E.random_points(curve, ctx.witness_size - first_slot)
    key = m.CommitKey(ctx, curve, E.points_to_bytes(gens), first_slot, window)
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    d_recs = torch.from_numpy(recs.view(np.int32)).to(dev)
    d_bodies = torch.zeros((n, ctx.body_bytes), dtype=torch.uint8, device=dev)
    d_pub = torch.zeros((n, ctx.public_words), dtype=torch.int32, device=dev)
    d_st = torch.zeros(n, dtype=torch.int32, device=dev)
    ctx.run_device(d_recs.data_ptr(), n, d_bodies.data_ptr(), 0, d_pub.data_ptr(), d_st.data_ptr(), s)
    want = torch.zeros((n, 64), dtype=torch.uint8, device=dev)
    cst = torch.zeros(n, dtype=torch.int32, device=dev)
    key.commit_device(d_bodies.data_ptr(), n, 0, want.data_ptr(), cst.data_ptr(), s)
    got = torch.full((n, 64), 7, dtype=torch.uint8, device=dev)
    pub2 = torch.zeros_like(d_pub)
    st2 = torch.full((n,), -1, dtype=torch.int32, device=dev)
    key.commit_records_device(d_recs.data_ptr(), n, got.data_ptr(), st2.data_ptr(), pub2.data_ptr(), s)
    torch.cuda.synchronize()
    st = d_st.cpu().numpy()
    assert torch.equal(st2, d_st) and (circuit == "compression" or st[7] != 0) and int((st != 0).sum()) == (0 if circuit == "compression" else 1)
    ok = torch.from_numpy(st == 0).to(dev)
    assert torch.equal(pub2[ok], d_pub[ok])
    assert torch.equal(got[ok], want[ok])
    assert int(got[~ok].abs().sum().item()) == 0               # rejected: the point at infinity
    hpts, hpub, hst = key.commit_records(recs)                # the host-buffer form
    assert np.array_equal(hpts, got.cpu().numpy()) and np.array_equal(hst, st) and np.array_equal(hpub.view(np.int32), pub2.cpu().numpy())
    probe = [0, n - 1] if circuit == "compression" else [0, 11, 12, n - 1]
    bad, bodies = T.oracle_batch_u32(circuit, recs[probe])
    assert bad == 0
    vals = _slot_values(bodies.copy())
    pts = got.cpu().numpy()
    for j, i in enumerate(probe):
        assert E.point_from_bytes(pts[i].tobytes()) == E.commit(vals[j][first_slot:], gens, curve), i
    key.close(); ctx.close()


def test_bodies_mode_takes_a_tabulated_inverse_only_when_the_slot_holds_it():
    """r03: bodies mode of the O2 nova builds adds ONE tabulated point (+-1/k) G per IsZero gadget too — k from the body's four
    input slots, and only if the body's inverse slot IS that scalar; a body that says anything else (an inverse off by one, another
    gadget's inverse, 0, inputs that do not fit the inverses, an input that is no word) must be committed to as it stands, through
    the windows.  Plain-integer MSM of every tampered body."""
    import torch
    m = T.pkg()
    W = T.workloads()
    circuit, curve, first_slot = "nova_vesta", "vesta", 5
    n = 7
    recs = W.config3_nova(n, first=31)
    bad, bodies = T.oracle_batch_u32(circuit, recs)
    assert bad == 0
    host = bodies.copy().reshape(n, -1, 32)
    # the input slots, found by their values among the first 48 (the O2 build merges pass-through outputs with their inputs: total_depth
    # and leaf_depth live among the outputs); leaf_depth = total_depth in this workload, so
    # "total_depth" is every slot that holds it
    def holds(col):
        return [s_ for s_ in range(1, 48) if all(int.from_bytes(host[i, s_].tobytes(), "little") == int(recs[i, col]) for i in range(n))]
    slot = {"n_blocks": holds(0), "block_count": holds(1), "total_depth": holds(13), "depth": holds(14)}
    assert all(len(v) >= 1 for v in slot.values()) and len(slot["depth"]) == 1 and len(slot["block_count"]) == 1, slot
    p = T.PRIME[circuit] if hasattr(T, "PRIME") else 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001
    rng = np.random.default_rng(5)
    def put(i, s_, v):
        host[i, s_] = np.frombuffer(int(v % (1 << 256)).to_bytes(32, "little"), dtype=np.uint8)
    def wide_slots(i):
        return [s_ for s_ in range(first_slot, host.shape[1]) if int.from_bytes(host[i, s_, 8:].tobytes(), "little")]
    for i in range(n):
        ws = wide_slots(i)
        assert len(ws) >= 60
        a, b = (int(x) for x in rng.choice(ws, 2, replace=False))
        va = int.from_bytes(host[i, a].tobytes(), "little")
        if i == 0: put(i, a, (va + 1) % p)                                        # an inverse off by one
        elif i == 1: put(i, a, int.from_bytes(host[i, b].tobytes(), "little"))   # another gadget's inverse
        elif i == 2: put(i, a, 0)                                                 # 0 where 1 / k stands
        elif i == 3: put(i, slot["depth"][0], int(recs[i, 14]) + 1)               # the inputs no longer fit the eqs[] inverses
        elif i == 4:
            for s_ in slot["total_depth"]: put(i, s_, (1 << 200) + 7)             # an input that is no word
        elif i == 5: put(i, slot["block_count"][0], int(recs[i, 1]) + 3)          # ... nor gadgets 1 and 2
        # i == 6: untouched
    flat = host.reshape(n, -1)
    vals = _slot_values(flat.copy())
    ctx = m.Context(circuit, 0)
    gens = E.random_points(curve, ctx.witness_size - first_slot)
    key = m.CommitKey(ctx, curve, E.points_to_bytes(gens), first_slot, 16)
    dev = torch.device("cuda:0")
    d = torch.from_numpy(flat.copy()).to(dev)
    pts = torch.zeros((n, 64), dtype=torch.uint8, device=dev)
    st = torch.zeros(n, dtype=torch.int32, device=dev)
    key.commit_device(d.data_ptr(), n, 0, pts.data_ptr(), st.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got, flags = pts.cpu().numpy(), st.cpu().numpy()
    for i in range(n):
        if i == 4:                                           # a 32-bit slot holding 2^200: outside the key's domain — flagged, as ever
            assert flags[i] != 0
            continue
        assert flags[i] == 0 and E.point_from_bytes(got[i].tobytes()) == E.commit(vals[i][first_slot:], gens, curve), i
    key.close(); ctx.close()


def test_inverse_tables_only_for_the_curve_whose_order_is_the_circuits_prime():
    """the tabulated (+-1/k) G stand for a slot that holds 1/k in the CIRCUIT's field; that is the scalar the windows
    add up to only on the curve whose group order is that prime.  A key that pairs the --prime vesta circuit with BN254 G1 (the C
    API allows it) must therefore go through the windows: bodies mode, records mode and the plain-integer sum agree."""
    import torch
    m = T.pkg()
    W = T.workloads()
    circuit, curve, first_slot = "nova_vesta", "bn254_g1", 22950          # the control block: all 67 inverses, few other slots
    n = 3
    recs = W.config3_nova(n, first=77)
    assert any(int(r[14]) > 0 for r in recs)                               # a negative argument (-depth) among them
    bad, bodies = T.oracle_batch_u32(circuit, recs)
    assert bad == 0
    vals = _slot_values(bodies.copy())
    ctx = m.Context(circuit, 0)
    gens = E.random_points(curve, ctx.witness_size - first_slot)
    key = m.CommitKey(ctx, curve, E.points_to_bytes(gens), first_slot, 12)
    b = m.Batch(ctx, n)
    b.run(recs)
    pts, st = b.commit(key)
    rpts, _, rst = key.commit_records(recs)
    assert (st == 0).all() and (rst == 0).all() and np.array_equal(pts, rpts)
    for i in range(n):
        assert E.point_from_bytes(pts[i].tobytes()) == E.commit(vals[i][first_slot:], gens, curve), i
    # b3w_commit_key_count: the kernel counts its mixed additions = the witnesses' non-zero 12-bit windows (no tabulated inverses here)
    key.count(True)
    b.commit(key)
    adds, wits = key.counts()
    key.count(False)
    want = 0
    for i in range(n):
        bits = []
        for s_, w in zip(range(first_slot, ctx.witness_size), ctx.slot_widths()[first_slot:]):
            bits += [(vals[i][s_] >> k) & 1 for k in range(int(w))]
        bits += [0] * (-len(bits) % 12)
        want += sum(1 for k in range(0, len(bits), 12) if any(bits[k:k + 12]))
    assert (adds, wits) == (want, n)
    key.close(); b.close(); ctx.close()


def test_chained_pass_commit_only_matches_the_commit_consumer():
    """b3w_chain_commit_only: the same points as committing every batch of bodies in the ring, and still BLAKE3(preimage)."""
    import torch, blake3_ref
    m = T.pkg()
    circuit, curve = "nova_vesta", "vesta"
    ctx = m.Context(circuit, 0)
    data = ((np.arange(8 * 1024, dtype=np.uint64) * 40503 + 11) % 253).astype(np.uint8)        # 8 chunks: 128 leaf + 24 parent steps
    key = m.CommitKey(ctx, curve, E.points_to_bytes(E.random_points(curve, T.NWIT[circuit])))
    dev = torch.device("cuda:0")
    a = torch.zeros((152, 64), dtype=torch.uint8, device=dev)
    b = torch.zeros((152, 64), dtype=torch.uint8, device=dev)

    def consumer(view, first, k):
        key.commit_device(view.data_ptr(), k, view.stride(0), a.data_ptr() + 64 * first, 0, torch.cuda.current_stream().cuda_stream)
    out = m.chain.fold_witnesses(ctx, data, batch_steps=48, ring=2, consumer=consumer)
    torch.cuda.synchronize()
    root = out["root"].cpu().numpy().view(np.uint32).tobytes()
    out2 = m.chain.fold_witnesses(ctx, data, batch_steps=48, ring=2, commit_only=(key, b))
    torch.cuda.synchronize()
    assert out2["root"].cpu().numpy().view(np.uint32).tobytes() == root == blake3_ref.blake3(data.tobytes())
    assert int(out2["status"].abs().sum().item()) == 0 and torch.equal(a, b) and int(b.max(dim=1).values.min().item()) > 0
    # b3w_chain_commit_from_records: the same points from the records WHILE the bodies are written and the consumer sees every batch
    # (here it commits to the bodies once more, into another array: three ways to the same 152 points)
    c = torch.zeros((152, 64), dtype=torch.uint8, device=dev)
    a.zero_()
    seen = []
    out3 = m.chain.fold_witnesses(ctx, data, batch_steps=48, ring=2, consumer=lambda v, f, k: (seen.append(k), consumer(v, f, k)), commit_records=(key, c))
    torch.cuda.synchronize()
    assert sum(seen) == 152 and torch.equal(c, b) and torch.equal(a, b) and out3["root"].cpu().numpy().view(np.uint32).tobytes() == root
    # b3w_chain_commit_overlap: wherever the commitments run, the points are the same; with the chain's own constraint check in
    # front of the consumer every step still satisfies the circuit.  GATED (what "auto" picks when something reads the batch) makes a
    # promise about order: what runs on the caller's stream after a batch's witness kernel starts when the batch's commitments are
    # done too — so a consumer that copies the batch's POINTS on that stream sees the finished points of exactly its batch.
    r1cs = m.R1cs(ctx)
    for mode in ("serial", "free", "gated", "auto"):
        for chk in (None, r1cs):
            c.zero_()
            snap = torch.zeros_like(c)

            def copy_points(view, first, k):
                snap[first:first + k].copy_(c[first:first + k])         # on the current (= the chain's caller) stream
            o = m.chain.fold_witnesses(ctx, data, batch_steps=48, ring=2, consumer=copy_points, commit_records=(key, c), check=chk, commit_overlap=mode)
            torch.cuda.synchronize()
            assert torch.equal(c, b), (mode, chk is not None)
            assert int(o["status"].abs().sum().item()) == 0 and o["root"].cpu().numpy().view(np.uint32).tobytes() == root
            if chk is not None:
                assert o["violations"].shape == (152,) and int(o["violations"].abs().sum().item()) == 0
            if mode in ("serial", "gated", "auto"):
                assert torch.equal(snap, b), f"{mode}: a consumer ran before its batch's commitments were complete"
    # a run without a consumer and without a check: auto = free; and back to bodies only
    c.zero_()
    m.chain.fold_witnesses(ctx, data, batch_steps=48, ring=2, commit_records=(key, c))
    torch.cuda.synchronize()
    assert torch.equal(c, b)
    assert m.lib().b3w_chain_commit_overlap(next(iter(ctx._chain_cache.values())), 7) == 100      # B3W_E_BAD_ARGUMENT
    o = m.chain.fold_witnesses(ctx, data, batch_steps=48, ring=2)
    torch.cuda.synchronize()
    assert o["violations"] is None and o["root"].cpu().numpy().view(np.uint32).tobytes() == root
    r1cs.close(); key.close(); ctx.close()


def test_commitments_from_records_across_a_chunk_border_and_argument_checks():
    """Records mode works in chunks of 32 768 witnesses: points on both sides of the border equal the commitments of
    the bodies of the same records; misaligned bodies are refused."""
    import torch
    m = T.pkg()
    n = 32768 + 900
    ctx = m.Context("compression", 0)
    recs = T.workloads().config2_compression(n, first=1)
    key = m.CommitKey(ctx, "bn254_g1", E.points_to_bytes(E.random_points("bn254_g1", ctx.witness_size)), 0, 12)
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    d_recs = torch.from_numpy(recs.view(np.int32)).to(dev)
    got = torch.zeros((n, 64), dtype=torch.uint8, device=dev)
    st = torch.full((n,), -1, dtype=torch.int32, device=dev)
    pub = torch.zeros((n, 16), dtype=torch.int32, device=dev)
    key.commit_records_device(d_recs.data_ptr(), n, got.data_ptr(), st.data_ptr(), pub.data_ptr(), s)
    sel = torch.tensor([0, 1, 32766, 32767, 32768, 32769, n - 2, n - 1], device=dev)
    k = sel.numel()
    d_sub = d_recs[sel].contiguous()
    d_bodies = torch.zeros((k, ctx.body_bytes), dtype=torch.uint8, device=dev)
    pub2 = torch.zeros((k, 16), dtype=torch.int32, device=dev)
    st2 = torch.zeros(k, dtype=torch.int32, device=dev)
    ctx.run_device(d_sub.data_ptr(), k, d_bodies.data_ptr(), 0, pub2.data_ptr(), st2.data_ptr(), s)
    want = torch.zeros((k, 64), dtype=torch.uint8, device=dev)
    key.commit_device(d_bodies.data_ptr(), k, 0, want.data_ptr(), 0, s)
    torch.cuda.synchronize()
    assert int(st.abs().sum().item()) == 0 and torch.equal(got[sel], want) and torch.equal(pub[sel], pub2)
    assert int(got.max(dim=1).values.min().item()) > 0                       # every point written
    with pytest.raises(m.B3WError):
        key.commit_device(d_bodies.data_ptr() + 4, k - 1, 0, want.data_ptr(), 0, s)
    key.close(); ctx.close()


def test_device_reproduces_published_known_answers():
    """The commit kernel against PUBLISHED constants, not just against tests/ec_ref.py: hand-made bodies whose only
    non-zero slots are w[0] = 1 and the 32-bit output slots w[1], w[2] (b3w_batch_commit_device does not care whether
    a body is a valid witness, only that bit slots hold bits).
      BN254 G1 (EIP-196): generators (G, G): 1*G + 2*G = 3G as printed in EIP-196 / py_ecc;
                          generators (-, Q, 2^32 Q): lo*Q + hi*2^32 Q = k*Q with go-ethereum's bn256ScalarMul vector
                          "chfast1" (k = 0x11138ce750fa15c2 -> S); generators (P1, P2): go-ethereum's bn256Add vector "chfast1";
      Pallas (pasta_curves): generator (-1, 2): 1*G + 5*G = 6G, and (q - 1) G = -G pins 6G through ec_ref's order test."""
    import torch
    m = T.pkg()
    H = lambda s: int(s, 16)
    dev = torch.device("cuda:0")

    def commit_body(circuit, curve, slot_vals, gens3):
        ctx = m.Context(circuit, 0)
        nwit = ctx.witness_size
        body = np.zeros(nwit * 32, dtype=np.uint8)
        for s, v in slot_vals.items():
            body[32 * s:32 * s + 32] = np.frombuffer(int(v).to_bytes(32, "little"), dtype=np.uint8)
        filler = E.random_points(curve, 1, seed=b"kat")[0]
        gens = list(gens3) + [filler] * (nwit - len(gens3))
        key = m.CommitKey(ctx, curve, E.points_to_bytes(gens), window=12)
        d_body = torch.from_numpy(body).to(dev)
        d_pts = torch.zeros(64, dtype=torch.uint8, device=dev)
        d_st = torch.full((1,), -1, dtype=torch.int32, device=dev)
        key.commit_device(d_body.data_ptr(), 1, 0, d_pts.data_ptr(), d_st.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert int(d_st.item()) == 0
        out = E.point_from_bytes(d_pts.cpu().numpy().tobytes())
        key.close(); ctx.close()
        return out

    q_bn, _ = E.CURVES["bn254_g1"]
    G = (1, 2)
    G3 = (3353031288059533942658390886683067124040920775575537747144343083137631628272,
          19321533766552368860946552437480515441416830039777911637913418824951667761761)
    assert commit_body("compression", "bn254_g1", {0: 1, 1: 2}, [G, G]) == G3
    # bn256ScalarMul "chfast1": k * Q = S, with k split over two 32-bit slots; then bn256Add "chfast1": P1 + P2 = R
    Q = (H("2bd3e6d0f3b142924f5ca7b49ce5b9d54c4703d7ae5648e61d02268b1a0a9fb7"), H("21611ce0a6af85915e2f1d70300909ce2e49dfad4a4619c8390cae66cefdb204"))
    k = H("11138ce750fa15c2")
    S = (H("070a8d6a982153cae4be29d434e8faef8a47b274a053f5a4ee2a6c9c13c31e5c"), H("031b8ce914eba3a9ffb989f9cdd5b0f01943074bf4f0f315690ec3cec6981afc"))
    Q32 = E.mul(1 << 32, Q, q_bn)
    O = commit_body("compression", "bn254_g1", {0: 0, 1: k & 0xFFFFFFFF, 2: k >> 32}, [G, Q, Q32])
    assert O == S
    P1 = (H("18b18acfb4c2c30276db5411368e7185b311dd124691610c5d3b74034e093dc9"), H("063c909c4720840cb5134cb9f59fa749755796819658d32efc0d288198f37266"))
    P2 = (H("07c2b7f58a84bd6145f00c9c2bc0bb1a187f20ff2c92963a88019e7c6a014eed"), H("06614e20c147e940f2d70da3f74c9a17df361706a4485c742bd6788478fa17d7"))
    R = (H("2243525c5efd4b9c3d3c45ac0ca3fe4dd85e830a4ce6b65fa1eeaee202839703"), H("301d1d33be6da8e509df21cc35964723180eed7532537db9ae5e7d48f195c915"))
    assert commit_body("compression", "bn254_g1", {0: 1, 1: 1}, [P1, P2]) == R
    # Pallas: generator (-1, 2)
    p_pa, _ = E.CURVES["pallas"]
    Gp = (p_pa - 1, 2)
    assert commit_body("nova_vesta", "pallas", {0: 1, 1: 5}, [Gp, Gp]) == E.mul(6, Gp, p_pa)
    assert commit_body("nova_vesta", "pallas", {0: 1}, [E.neg(Gp, p_pa)]) == E.mul(T.VESTA_Q - 1, Gp, p_pa)


@pytest.mark.parametrize("circuit,curve,first_slot,window", [("compression", "bn254_g1", 0, 16), ("compression", "bn254_g1", 45, 12),
                                                             ("nova_vesta", "vesta", 0, 16), ("nova_bn254", "bn254_g1", 1, 12),
                                                             ("nova_bn254_o1", "bn254_g1", 3, 12)])
def test_folded_keys_commit_to_the_same_points(circuit, curve, first_slot, window):
    """include/b3wit.h "FOLDED keys": the slots that the circuit's linear constraints express through others (every 32-bit
    word through its bits; fold.py derives that from the .r1cs image alone) are folded into those slots' generators and drop
    out of the table.  The commitment of every witness must be the same point as under the unfolded key — from the bodies and
    straight from the records, rejected nova steps included — with about half the virtual slots for the O1-style builds
    (blake3_compression, the circomkit nova build) and three quarters for the O2 builds."""
    import torch
    m = T.pkg()
    W = T.workloads()
    n = 200
    recs = W.config2_compression(n, first=31) if circuit == "compression" else W.config3_nova(n, first=31)
    if circuit != "compression":
        recs = recs.copy()
        recs[5, 14] = recs[5, 12]                             # a rejected step
    ctx = m.Context(circuit, 0)
    gens = E.points_to_bytes(E.random_points(curve, ctx.witness_size - first_slot))
    plain = m.CommitKey(ctx, curve, gens, first_slot, window)
    folded = m.CommitKey(ctx, curve, gens, first_slot, window, fold=True)
    st = folded.fold_stats
    # compression: every word folds into its bits (53 457 -> 23 377 virtual slots); the circomkit nova build keeps 190 full field
    # elements of 256 virtual slots each; the O2 builds have no linear constraints left, but 460 of their 643 words are
    # "31 bit slots + 2^i * one more bit" (the booleanity of the bit circom took out): those words shrink to that one bit
    assert folded.folded_slots >= 400 and st["virtual_slots_folded"] < (0.62 if circuit == "compression" else 0.78) * st["virtual_slots"], st
    assert (st["single_bit_words"] >= 400) == (circuit in ("nova_vesta", "nova_bn254")), st
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    d_recs = torch.from_numpy(recs.view(np.int32)).to(dev)
    d_bodies = torch.zeros((n, ctx.body_bytes), dtype=torch.uint8, device=dev)
    d_st = torch.zeros(n, dtype=torch.int32, device=dev)
    ctx.run_device(d_recs.data_ptr(), n, d_bodies.data_ptr(), 0, 0, d_st.data_ptr(), s)
    out = {}
    for name, key in (("plain", plain), ("folded", folded)):
        a = torch.full((n, 64), 7, dtype=torch.uint8, device=dev)
        b = torch.full((n, 64), 7, dtype=torch.uint8, device=dev)
        sa = torch.full((n,), -1, dtype=torch.int32, device=dev)
        sb = torch.full((n,), -1, dtype=torch.int32, device=dev)
        key.commit_device(d_bodies.data_ptr(), n, 0, a.data_ptr(), sa.data_ptr(), s)
        key.commit_records_device(d_recs.data_ptr(), n, b.data_ptr(), sb.data_ptr(), 0, s)
        torch.cuda.synchronize()
        out[name] = (a.cpu().numpy(), b.cpu().numpy(), sa.cpu().numpy(), sb.cpu().numpy())
    ok = d_st.cpu().numpy() == 0
    assert int((~ok).sum()) == (0 if circuit == "compression" else 1)
    assert np.array_equal(out["plain"][0][ok], out["folded"][0][ok]), "bodies: folded key gives other points"
    assert np.array_equal(out["plain"][1], out["folded"][1]), "records: folded key gives other points"
    assert np.array_equal(out["plain"][0][ok], out["plain"][1][ok])
    assert (out["folded"][2][ok] == 0).all() and np.array_equal(out["folded"][3], out["plain"][3])
    assert int(out["folded"][1][~ok].astype(np.int64).sum()) == 0               # rejected: the point at infinity
    plain.close(); folded.close(); ctx.close()


def test_commit_only_pass_makes_its_points_at_the_end_of_a_run_call():
    """r05: a commit-only chained pass leaves its batches' projective sums and makes the points by ONE launch per run call (four
    witnesses share an inversion: b3w_commit_normalize_many_kernel).  Step counts that are no multiples of four, batches that are no
    multiples of anything, and a key that commits nothing (every sum is the point at infinity -> (0, 0)): the points equal those of
    b3w_commit_records on the pass's own records, which normalises witness by witness."""
    import torch
    m = T.pkg()
    circuit, curve = "nova_vesta", "vesta"
    ctx = m.Context(circuit, 0)
    dev = torch.device("cuda:0")
    data = ((np.arange(2 * 1024 + 100, dtype=np.uint64) * 2654435761 + 5) % 251).astype(np.uint8)      # 16 + 16 + 2 leaf steps
    gens = E.points_to_bytes(E.random_points(curve, T.NWIT[circuit]))
    for key_bytes, what in ((gens, "real"), (bytes(len(gens)), "nothing committed")):
        key = m.CommitKey(ctx, curve, key_bytes, window=12)
        probe = m.chain.fold_witnesses(ctx, data, batch_steps=7, ring=2)
        rows = probe["n_leaf_steps"] + probe["n_parent_steps"]
        assert probe["n_leaf_steps"] == 34 and (probe["n_leaf_steps"] % 4 != 0 or probe["n_parent_steps"] % 4 != 0)
        pts = torch.full((rows, 64), 0xAB, dtype=torch.uint8, device=dev)
        out = m.chain.fold_witnesses(ctx, data, batch_steps=7, ring=2, commit_only=(key, pts))
        torch.cuda.synchronize()
        assert int(out["status"].abs().sum().item()) == 0
        recs = out["records"].cpu().numpy().astype(np.uint32)
        want, _, st = key.commit_records(recs)
        assert int(np.abs(st).sum()) == 0
        got = pts.cpu().numpy()
        assert want.shape == got.shape and np.array_equal(got, want), what
        if what != "real":
            assert int(got.max()) == 0
        else:
            assert int(got.max(axis=1).min()) > 0
        key.close()
    ctx.close()


@pytest.mark.parametrize("use_null_stream", [True, False])
def test_gated_commitments_of_a_many_slice_pass_equal_the_commit_only_points(use_null_stream):
    """r05: under GATED the TRACE images of a batch are written on the CALLER's stream in front of its witness kernel and the commit kernel
    waits for them on the commit stream — ordered by an event, also when the caller's stream is the null stream (a first version took
    "null" for "no stream", lost the ordering and read records the planner had not written yet: only a pass of several slices shows it).
    2 MiB = 32 768 leaf steps in batches of 4 096, planned slice by slice: the points of check + commit-from-records equal those of the
    commit-only pass, every step satisfies the circuit, the fold ends in BLAKE3(preimage)."""
    import torch, blake3_ref
    m = T.pkg()
    ctx = m.Context("nova_vesta", 0)
    dev = torch.device("cuda:0")
    data = m.workloads.lcg_preimage(2 << 20, seed=9)
    key = m.CommitKey(ctx, "pallas", E.points_to_bytes(E.random_points("pallas", T.NWIT["nova_vesta"])), window=12)
    r1cs = m.R1cs(ctx)
    probe = m.chain.fold_witnesses(ctx, data, batch_steps=4096, ring=2)
    rows = probe["n_leaf_steps"] + probe["n_parent_steps"]
    want = torch.zeros((rows, 64), dtype=torch.uint8, device=dev)
    m.chain.fold_witnesses(ctx, data, batch_steps=4096, ring=2, commit_only=(key, want))
    torch.cuda.synchronize()
    got = torch.zeros((rows, 64), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream() if use_null_stream else torch.cuda.Stream()
    assert (st.cuda_stream == 0) == use_null_stream
    with torch.cuda.stream(st):
        for _ in range(2):                                     # (twice: the second pass finds every buffer in place and runs ahead of the host)
            got.zero_()
            out = m.chain.fold_witnesses(ctx, data, batch_steps=4096, ring=2, commit_records=(key, got), check=r1cs, commit_overlap="gated")
        st.synchronize()
    assert torch.equal(got, want) and int(got.max(dim=1).values.min().item()) > 0
    assert int(out["status"].abs().sum().item()) == 0 and int(out["violations"].abs().sum().item()) == 0
    assert out["root"].cpu().numpy().view(np.uint32).tobytes() == blake3_ref.blake3(data.tobytes())
    r1cs.close(); key.close(); ctx.close()


def test_automatic_window_falls_back_when_its_table_does_not_fit(monkeypatch):
    """the automatic window is a guess from ONE reading of the free memory; when the 18-bit table then fails to allocate
    (other ranks sizing their keys on the same GPU, torch or the placement pool in between, fragmentation) the key must come out with 16
    or 12 bits instead of B3W_E_HIP.  Here: the choice is told to believe in memory that a torch tensor already holds."""
    import torch
    m = T.pkg()
    ctx = m.Context("nova_vesta", 0)
    gens = E.points_to_bytes(E.random_points("vesta", ctx.witness_size))
    torch.cuda.empty_cache()
    m.lib().b3w_bodies_trim()
    free, total = torch.cuda.mem_get_info(0)
    # an unfolded nova key: 51 GB at 18 bits, 13 GB at 16.  Leave 30 GB: too little for the first, enough for the second.
    hog = torch.empty(max(free - (30 << 30), 1 << 20), dtype=torch.uint8, device="cuda:0")
    monkeypatch.setenv("B3W_COMMIT_ASSUME_FREE_GIB", "1000")
    key = m.CommitKey(ctx, "vesta", gens, 0, 0)
    assert key.window == 16, key.window
    # an explicit window that does not fit is the caller's decision: an error, no fallback
    with pytest.raises(m.B3WError):
        m.CommitKey(ctx, "vesta", gens, 0, 18)
    torch.cuda.synchronize()
    # the key that came out works: same points as an explicit 16-bit key
    n = 8
    recs = T.workloads().config3_nova(n)
    d_recs = torch.from_numpy(recs.view(np.int32)).to("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    monkeypatch.delenv("B3W_COMMIT_ASSUME_FREE_GIB")
    del hog
    torch.cuda.empty_cache()
    ref = m.CommitKey(ctx, "vesta", gens, 0, 16)
    out = []
    for k in (key, ref):
        p = torch.zeros((n, 64), dtype=torch.uint8, device="cuda:0")
        st = torch.zeros(n, dtype=torch.int32, device="cuda:0")
        k.commit_records_device(d_recs.data_ptr(), n, p.data_ptr(), st.data_ptr(), 0, s)
        torch.cuda.synchronize()
        out.append(p.cpu().numpy())
    assert np.array_equal(out[0], out[1]) and int(out[0].max(axis=1).min()) > 0
    key.close(); ref.close(); ctx.close()
