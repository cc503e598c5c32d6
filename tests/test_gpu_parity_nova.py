"""Parity of the nova step kernels (BN254 O2, Vesta O2, BN254 circomkit/O1) against the oracle and
the WASM-generated goldens, through the C-ABI.  Bit-exact."""
import os
import numpy as np
import pytest
import b3w_testlib as T

pytestmark = pytest.mark.gpu
NOVA = ["nova_bn254", "nova_vesta", "nova_bn254_o1"]


@pytest.fixture(scope="module")
def m():
    return T.pkg()


def _pub_from_body(body):
    return body.reshape(-1, 32)[1:16, :4].copy().view(np.uint32).reshape(15)


@pytest.mark.parametrize("circuit", NOVA)
def test_nova_batch_matches_oracle(m, circuit):
    recs = T.workloads().config3_nova(130, first=7)
    bad, want = T.oracle_batch_u32(circuit, recs)
    assert bad == 0
    want = want.copy()
    for variant in (0, 1, 22, 27, 36, 84) + ((2, 3, 4, 200, 201) if circuit != "nova_bn254_o1" else ()):   # 20 + s: SLICED (any s in 2..64); 4: persistent grid; 200 / 201: fill-ordered
        os.environ["B3W_VARIANT"] = str(variant)
        try:
            ctx = m.Context(circuit, 0)
        finally:
            del os.environ["B3W_VARIANT"]
        b = m.Batch(ctx, 130)
        b.run(recs)
        pub, st = b.outputs()
        assert (st == 0).all()
        for i in range(recs.shape[0]):
            got = b.fetch(i)
            assert np.array_equal(got, want[i]), (circuit, variant, i, np.nonzero(got != want[i])[0][:8] // 32)
            assert np.array_equal(pub[i], _pub_from_body(want[i]))
        b.close(); ctx.close()


@pytest.mark.parametrize("circuit", NOVA)
def test_nova_goldens_through_witness_calculator(m, circuit):
    g = T.golden(circuit)
    wc = m.builder(circuit)
    assert wc.witnessSize == T.NWIT[circuit] and wc.prime == T.PRIME[circuit]
    nok = nassert = nwild = 0
    for case in g["cases"]:
        canonical = T.is_canonical_u32(circuit, case["input"])
        if "error" in case:
            with pytest.raises(m.B3WError, match="Assert Failed") as e:
                wc.calculateWTNSBin(case["input"], 0)
            assert e.value.status == m.B3W_E_ASSERT_FAILED
            assert str(e.value) == case["error"], case["name"]      # the WASM's own trace, line for line
            nassert += 1
        else:
            img = wc.calculateWTNSBin(case["input"], 0)   # non-canonical inputs go through the exact device kernel
            assert T.sha256(img) == case["wtns_sha256"], case["name"]
            nok += 1
            nwild += 0 if canonical else 1
    assert nok >= 80 and nassert >= 6 and nwild >= 4
    for f in os.listdir(T.GOLD):
        if f.startswith(circuit + ".") and f.endswith(".wtns.gz"):
            case = next(c for c in g["cases"] if c["name"] == f[len(circuit) + 1:-len(".wtns.gz")])
            assert wc.calculateWTNSBin(case["input"], 0).tobytes() == T.golden_image(f)


@pytest.mark.parametrize("circuit", ["nova_vesta", "nova_bn254_o1"])
def test_nova_mixed_valid_and_rejected_steps(m, circuit):
    """Steps the circuit rejects (depth >= leaf_depth, 9-bit comparator overflow) report status 4 and leave
    their body untouched; neighbours in the same wave are unaffected."""
    recs = T.workloads().config3_nova(37, first=500).copy()
    bad_idx = [0, 5, 6, 18, 36]
    recs[0, 14] = recs[0, 12]            # depth == leaf_depth
    recs[5, 14] = recs[5, 12] + 3        # depth > leaf_depth
    recs[6, 12] = recs[6, 14] + 700      # leaf_depth far above depth: LessThan(8) input needs > 9 bits
    recs[18, 12] = 0; recs[18, 14] = 0   # 0,0
    recs[36, 14] = 4000000000            # depth far above leaf_depth
    nbad, want = T.oracle_batch_u32(circuit, recs)
    assert nbad == len(bad_idx)
    want = want.copy()
    ctx = m.Context(circuit, 0)
    b = m.Batch(ctx, 37)
    b.run(np.zeros((37, 32), np.uint32) + np.uint32(1))      # dirty the buffers with an all-rejected batch first
    b.run(recs)
    pub, st = b.outputs()
    assert [i for i in range(37) if st[i] != 0] == bad_idx and all(st[i] == 4 for i in bad_idx)
    for i in range(37):
        if i not in bad_idx:
            assert np.array_equal(b.fetch(i), want[i]), i
    b.close(); ctx.close()


def test_nova_large_isZero_arguments_take_the_general_inverse_path(m):
    """block_count / depth far outside the small-inverse table: (p*t+1)/k path of the kernel."""
    recs = T.workloads().config3_nova(24, first=900).copy()
    leaf = recs[:, 14] == recs[:, 12] - 1
    recs[leaf, 1] = np.arange(leaf.sum(), dtype=np.uint32) * np.uint32(178956970) + np.uint32(5000)   # block_count
    recs[:, 13] = recs[:, 13] + np.uint32(100000)                                                       # total_depth: eq args ~1e5
    for circuit in ("nova_bn254", "nova_vesta", "nova_bn254_o1"):
        nbad, want = T.oracle_batch_u32(circuit, recs)
        assert nbad == 0
        want = want.copy()
        ctx = m.Context(circuit, 0)
        b = m.Batch(ctx, 24)
        b.run(recs)
        _, st = b.outputs()
        assert (st == 0).all()
        for i in range(24):
            assert np.array_equal(b.fetch(i), want[i]), (circuit, i)
        b.close(); ctx.close()


def _nova_public_np(recs):
    """Independent numpy model of the step's public outputs (h_out via plain BLAKE3)."""
    from test_gpu_parity import _blake3_compress_np
    IV = T.workloads().IV.astype(np.uint32)
    nb, bc, h = recs[:, 0], recs[:, 1], recs[:, 2:10]
    cil, cih, ld, td, depth, mm, b = recs[:, 10], recs[:, 11], recs[:, 12], recs[:, 13], recs[:, 14], recs[:, 15:31], recs[:, 31]
    parent = depth.astype(np.int64) < ld.astype(np.int64) - 1
    root = depth == 0
    e0, e1 = bc == 0, nb.astype(np.int64) - 1 == bc.astype(np.int64)
    first, last = e0 & ~parent, e1 & ~parent
    d = first.astype(np.uint32) + 2 * last.astype(np.uint32) + 8 * ((parent | e1) & root).astype(np.uint32) + 4 * parent.astype(np.uint32)
    istar = td.astype(np.int64) - 2 - depth.astype(np.int64)
    ci = cil.astype(np.uint64) | (cih.astype(np.uint64) << np.uint64(32))
    ok = (istar >= 0) & (istar < 64)
    bit = (ci >> np.clip(istar, 0, 63).astype(np.uint64)) & np.uint64(1)
    dl = np.where(parent, ok & (bit == 0), True)
    msg = mm.copy()
    left = np.where(dl[:, None], h, mm[:, :8])
    right = np.where(dl[:, None], mm[:, :8], h)
    msg[parent] = np.concatenate([left, right], axis=1)[parent]
    hc = np.where(parent[:, None], IV[None, :], h)
    t0, t1 = np.where(parent, 0, cil).astype(np.uint32), np.where(parent, 0, cih).astype(np.uint32)
    out = _blake3_compress_np(hc.astype(np.uint32), msg.astype(np.uint32), t0, t1, b.copy(), d.astype(np.uint32))
    bco = bc + (~parent).astype(np.uint32)
    dout = depth - (((last | parent) & ~root)).astype(np.uint32)
    return np.concatenate([nb[:, None], bco[:, None], out[:, :8], td[:, None], dout[:, None], cil[:, None], cih[:, None], ld[:, None]], axis=1).astype(np.uint32)


def test_nova_full_config3_batch(m):
    """BASELINE config 3 at full size: 65 536 Vesta steps (48.8 GB of witness bodies in HBM).  All public
    outputs against an independent numpy model; every body against the circuit's rank-1 constraints on the device; 4 096 bodies
    (every position inside a wave, first and last wave) byte-for-byte against the oracle."""
    import torch
    n = 65536
    recs = T.workloads().config3_nova(n)
    ctx = m.Context("nova_vesta", 0)
    dev = torch.device("cuda:0")
    free, _ = torch.cuda.mem_get_info()
    if free < n * ctx.body_bytes + (2 << 30):
        pytest.skip("not enough free HBM for the full batch")
    d_recs = torch.from_numpy(recs.view(np.int32)).to(dev)
    d_bodies = torch.empty((n, ctx.body_bytes), dtype=torch.uint8, device=dev)
    d_pub = torch.zeros((n, 15), dtype=torch.int32, device=dev)
    d_st = torch.full((n,), -1, dtype=torch.int32, device=dev)
    ctx.run_device(d_recs.data_ptr(), n, d_bodies.data_ptr(), 0, d_pub.data_ptr(), d_st.data_ptr(),
                   torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert (d_st == 0).all().item()
    pub = d_pub.cpu().numpy().view(np.uint32)
    assert np.array_equal(pub, _nova_public_np(recs))
    # slot 0 is the constant 1 in every body
    assert (d_bodies[:, :32].to(torch.int32).sum(dim=1) == 1).all().item() and (d_bodies[:, 0] == 1).all().item()
    # EVERY body against the step circuit's rank-1 constraints on the device (r05: the check that is independent of the witness
    # kernels, their layouts and the oracle — the derived system of the Vesta O2 build; 8 ms for the 65 536 bodies)
    r1cs = m.R1cs(ctx)
    d_viol = torch.full((n,), -1, dtype=torch.int32, device=dev)
    r1cs.check_device(d_bodies.data_ptr(), n, 0, d_viol.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert int(d_viol.abs().sum().item()) == 0, "a body of the full config-3 batch violates the step circuit's constraints"
    r1cs.close()
    # 4 096 bodies byte for byte against the oracle: the first and the last wave whole (a wave takes 8 bodies), and 510 random
    # bodies of every position w = i mod 8 inside a wave
    rng = np.random.default_rng(5)
    idx = [np.arange(8), np.arange(n - 8, n)] + [8 + 8 * rng.choice((n - 16) // 8, 510, replace=False) + w for w in range(8)]
    idx = np.sort(np.concatenate(idx))
    assert idx.size == 4096 and np.unique(idx).size == 4096 and {int(i) % 8 for i in idx} == set(range(8))
    for c0 in range(0, idx.size, 512):
        sel = idx[c0:c0 + 512]
        _, want = T.oracle_batch_u32("nova_vesta", recs[sel])
        got = d_bodies[torch.from_numpy(sel).to(dev)].cpu().numpy()
        assert np.array_equal(got, want), c0
    del d_bodies
    ctx.close()


def test_nova_persistent_grid_takes_several_groups_per_wave_with_rejected_steps(m):
    """Variant 4 of the O2 kernels (8 bodies a wave on a persistent grid of 512 waves — the default for batches of more than 32 768 steps):
    9 000 steps are more than two rounds of the grid, the last one ragged; rejected steps in the first, a middle and the last group must
    leave their bodies alone and must not leak their flags into the wave's next group.  Status and public outputs of every step and the
    bodies around every rejected step equal variant 3's; 300 bodies byte for byte against the oracle."""
    import torch
    n = 9000
    recs = T.workloads().config3_nova(n, first=3).copy()
    bad_idx = [0, 5, 4095, 4096, 4100, 8191, 8192, 8999]
    for i in bad_idx:
        recs[i, 14] = recs[i, 12]                             # depth >= leaf_depth: CheckDepth rejects the step
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    d_recs = torch.from_numpy(recs.view(np.int32)).to(dev)
    out = {}
    for variant in (3, 4):
        os.environ["B3W_VARIANT"] = str(variant)
        try:
            ctx = m.Context("nova_vesta", 0)
        finally:
            del os.environ["B3W_VARIANT"]
        d_bodies = torch.full((n, ctx.body_bytes), 0x6B, dtype=torch.uint8, device=dev)
        d_pub = torch.zeros((n, 15), dtype=torch.int32, device=dev)
        d_st = torch.full((n,), -1, dtype=torch.int32, device=dev)
        ctx.run_device(d_recs.data_ptr(), n, d_bodies.data_ptr(), 0, d_pub.data_ptr(), d_st.data_ptr(), s)
        torch.cuda.synchronize()
        out[variant] = (d_bodies, d_pub.cpu().numpy(), d_st.cpu().numpy())
        ctx.close()
    st = out[4][2]
    assert [i for i in range(n) if st[i] != 0] == bad_idx and np.array_equal(st, out[3][2])
    ok = st == 0
    assert np.array_equal(out[4][1][ok], out[3][1][ok])
    for i in bad_idx:
        assert bool((out[4][0][i] == 0x6B).all().item()), i                                  # a rejected step's body is left alone
    near = sorted({j for i in bad_idx for j in range(max(0, i - 9), min(n, i + 10))} - set(bad_idx))
    sel = torch.tensor(near, device=dev)
    assert torch.equal(out[4][0][sel], out[3][0][sel])
    idx = np.sort(np.random.default_rng(11).choice(np.flatnonzero(ok), 300, replace=False))
    _, want = T.oracle_batch_u32("nova_vesta", recs[idx])
    assert np.array_equal(out[4][0][torch.from_numpy(idx).to(dev)].cpu().numpy(), want)
