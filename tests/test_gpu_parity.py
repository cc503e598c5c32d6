"""Parity of the HIP path against the oracle and the committed goldens, through the C-ABI.
Bit-exact (integer work): every comparison is byte equality."""
import ctypes, os

import numpy as np
import pytest
import b3w_testlib as T

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def m():
    return T.pkg()


def _ctx(m, circuit):
    return m.Context(circuit, 0)


def _run_batch(m, ctx, recs, pitch=0):
    b = m.Batch(ctx, max(1, recs.shape[0]), pitch)
    b.run(recs)
    return b


def _blake3_compress_np(h, mm, t0, t1, b, d):
    """Independent plain BLAKE3 compression, vectorised over the batch (BLAKE3 spec section 2.2)."""
    IV = T.workloads().IV.astype(np.uint32)
    perm = [2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8]
    n = h.shape[0]
    v = [h[:, i].copy() for i in range(8)] + [np.full(n, IV[i], np.uint32) for i in range(4)] + [t0.copy(), t1.copy(), b.copy(), d.copy()]
    msg = [mm[:, i].copy() for i in range(16)]
    rot = lambda x, r: (x >> np.uint32(r)) | (x << np.uint32(32 - r))

    def g(a, b_, c, d_, x, y):
        v[a] = v[a] + v[b_] + x; v[d_] = rot(v[d_] ^ v[a], 16)
        v[c] = v[c] + v[d_]; v[b_] = rot(v[b_] ^ v[c], 12)
        v[a] = v[a] + v[b_] + y; v[d_] = rot(v[d_] ^ v[a], 8)
        v[c] = v[c] + v[d_]; v[b_] = rot(v[b_] ^ v[c], 7)
    for r in range(7):
        g(0, 4, 8, 12, msg[0], msg[1]); g(1, 5, 9, 13, msg[2], msg[3]); g(2, 6, 10, 14, msg[4], msg[5]); g(3, 7, 11, 15, msg[6], msg[7])
        g(0, 5, 10, 15, msg[8], msg[9]); g(1, 6, 11, 12, msg[10], msg[11]); g(2, 7, 8, 13, msg[12], msg[13]); g(3, 4, 9, 14, msg[14], msg[15])
        msg = [msg[perm[j]] for j in range(16)]
    out = [v[i] ^ v[i + 8] for i in range(8)] + [v[i + 8] ^ h[:, i] for i in range(8)]
    return np.stack(out, axis=1)


def test_compression_batch_matches_oracle_every_variant(m):
    recs = T.workloads().config2_compression(96)
    bad, want = T.oracle_batch_u32("compression", recs)
    assert bad == 0
    want = want.copy()
    for variant in list(range(9)) + [22, 24, 28, 36, 52, 84, 200, 201]:                      # 20 + s: SLICED, s waves per body; 200 / 201: fill-ordered
        os.environ["B3W_VARIANT"] = str(variant)
        try:
            ctx = _ctx(m, "compression")
        finally:
            del os.environ["B3W_VARIANT"]
        b = _run_batch(m, ctx, recs)
        pub, st = b.outputs()
        assert (st == 0).all()
        for i in range(recs.shape[0]):
            got = b.fetch(i)
            assert np.array_equal(got, want[i]), (variant, i, np.nonzero(got != want[i])[0][:8])
        assert np.array_equal(pub, want.reshape(96, -1, 32)[:, 1:17, :4].copy().view(np.uint32).reshape(96, 16))
        b.close(); ctx.close()


@pytest.mark.parametrize("n", [1, 2, 3, 5, 17, 63])
def test_compression_ragged_batches_and_padded_pitch(m, n):
    recs = T.workloads().config2_compression(n, first=1000)
    _, want = T.oracle_batch_u32("compression", recs)
    ctx = _ctx(m, "compression")
    for pitch in (0, 771072, 770976 + 32):           # contiguous, 128-B aligned, minimally padded
        b = _run_batch(m, ctx, recs, pitch)
        for i in range(n):
            assert np.array_equal(b.fetch(i), want[i])
        b.close()
    ctx.close()


def test_compression_edge_inputs(m):
    z = np.zeros((1, 28), np.uint32)
    recs = np.concatenate([z, z + np.uint32(0xFFFFFFFF), T.workloads().config1_cases()])
    _, want = T.oracle_batch_u32("compression", recs)
    ctx = _ctx(m, "compression")
    b = _run_batch(m, ctx, recs)
    for i in range(recs.shape[0]):
        assert np.array_equal(b.fetch(i), want[i])
    # config 1: the reference's own committed witness, byte for byte
    ref = T.golden_image("reference_testInp_witness.wtns.gz")
    assert ctx.wtns_header() + b.fetch(2).tobytes() == ref
    b.close(); ctx.close()


def test_witness_calculator_surface_against_goldens(m):
    """builder -> calculateWTNSBin / calculateBinWitness / calculateWitness as generate_witness.js uses them."""
    g = T.golden("compression")
    wc = m.builder("compression")
    assert wc.witnessSize == 24093 and wc.n32 == 8 and wc.prime == T.BN254_R and wc.circom_version() == 2
    nok = nerr = 0
    for case in g["cases"]:
        if "error" in case:
            # rejected by the circuit (also for non-canonical inputs: evaluated by the exact device kernel)
            with pytest.raises(m.B3WError, match="Error: Assert Failed.") as e:
                wc.calculateWTNSBin(case["input"], 0)
            assert e.value.status == m.B3W_E_ASSERT_FAILED
            assert str(e.value) == case["error"], case["name"]      # the WASM's own trace, line for line
            nerr += 1
            continue
        img = wc.calculateWTNSBin(case["input"], 0)      # includes negative / >= 2^32 message words
        assert T.sha256(img) == case["wtns_sha256"], case["name"]
        nok += 1
    assert nerr >= 6
    assert nok >= 45
    case = g["cases"][0]
    w = wc.calculateWitness(case["input"], 0)
    assert len(w) == 24093 and w[0] == 1 and [str(x) for x in w[:16]] == case["first16"]
    assert T.sha256(wc.calculateBinWitness(case["input"], 0)) == case["body_sha256"]
    img = wc.calculateWTNSBin(case["input"], 0).tobytes()
    assert img == T.golden_image("compression.config1_testInp.wtns.gz")


def test_witness_calculator_input_errors(m):
    # exact strings of witness_calculator.js:142-150,166-168 (probed on the reference WASM)
    wc = m.builder("compression")
    base = T.golden("compression")["cases"][0]["input"]
    inp = dict(base); del inp["b"]
    with pytest.raises(m.B3WError, match="Not all inputs have been set. Only 27 out of 28"):
        wc.calculateWitness(inp)
    inp = dict(base); inp["m"] = base["m"][:15]
    with pytest.raises(m.B3WError, match="Not enough values for input signal m\n"):
        wc.calculateWitness(inp)
    inp = dict(base); inp["m"] = base["m"] + [1]
    with pytest.raises(m.B3WError, match="Too many values for input signal m\n"):
        wc.calculateWitness(inp)
    inp = dict(base); inp["zz"] = 1
    with pytest.raises(m.B3WError, match="Too many values for input signal zz\n"):
        wc.calculateWitness(inp)
    # nested arrays, strings and key order do not matter
    inp = {"d": "0", "b": "0x40", "t": [[0], [0]], "m": [base["m"][:8], base["m"][8:]], "h": [str(x) for x in base["h"]]}
    assert T.sha256(wc.calculateBinWitness(inp)) == T.golden("compression")["cases"][0]["body_sha256"]
    # the calculator stays usable after a throw
    assert wc.calculateWitness(base)[0] == 1


def test_compression_full_config2_batch(m):
    """BASELINE config 2 at full size: 4096 witnesses, every byte checked against the oracle in chunks,
    plus the size-independent properties (outputs == plain BLAKE3, idempotence)."""
    import torch
    n = 4096
    W = T.workloads()
    recs = W.config2_compression(n)
    ctx = _ctx(m, "compression")
    dev = torch.device("cuda:0")
    d_recs = torch.from_numpy(recs.view(np.int32)).to(dev)
    d_bodies = torch.empty((n, ctx.body_bytes), dtype=torch.uint8, device=dev)
    d_pub = torch.zeros((n, 16), dtype=torch.int32, device=dev)
    d_st = torch.full((n,), -1, dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    ctx.run_device(d_recs.data_ptr(), n, d_bodies.data_ptr(), 0, d_pub.data_ptr(), d_st.data_ptr(), s)
    torch.cuda.synchronize()
    assert (d_st == 0).all().item()
    pub = d_pub.cpu().numpy().view(np.uint32)
    want_pub = _blake3_compress_np(recs[:, 0:8], recs[:, 8:24], recs[:, 24], recs[:, 25], recs[:, 26], recs[:, 27])
    assert np.array_equal(pub, want_pub)
    first = d_bodies.cpu().numpy() if False else None
    chunk = 256
    for c0 in range(0, n, chunk):
        _, want = T.oracle_batch_u32("compression", recs[c0:c0 + chunk])
        got = d_bodies[c0:c0 + chunk].cpu().numpy()
        assert np.array_equal(got, want), c0
    # idempotence: a second launch into a dirtied buffer reproduces the same bytes
    ck = d_bodies.view(torch.int64).sum().item()
    d_bodies.fill_(0xAB)
    ctx.run_device(d_recs.data_ptr(), n, d_bodies.data_ptr(), 0, d_pub.data_ptr(), d_st.data_ptr(), s)
    torch.cuda.synchronize()
    assert d_bodies.view(torch.int64).sum().item() == ck
    ctx.close()


def test_streaming_wtns_writer(m, tmp_path):
    """b3w_batch_write_wtns: every file equals calculateWTNSBin's image (goldens + oracle), rejected steps skipped."""
    import hashlib
    g = T.golden("compression")
    cases = [c for c in g["cases"] if "error" not in c and T.is_canonical_u32("compression", c["input"])][:40]
    W = T.workloads()
    recs = np.array([W.input_to_values(c["input"], W.COMPRESSION_KEYS) for c in cases], dtype=np.uint32)
    recs = np.concatenate([recs, W.config2_compression(260, first=9000)])          # three staging chunks (128 + 128 + 44): a buffer is reused
    ctx = _ctx(m, "compression")
    b = m.Batch(ctx, recs.shape[0], 771072)                                        # padded pitch on the device, files stay packed
    b.run(recs)
    _, want = T.oracle_batch_u32("compression", recs[40:])
    hdr = T.oracle_header("compression")
    for threads, prefix in ((0, "w"), (1, "one_"), (5, "five_")):                  # the library's choice, a single writer, a small pool
        assert b.write_wtns(tmp_path, prefix, threads=threads) == recs.shape[0]
        for i, c in enumerate(cases):
            assert hashlib.sha256((tmp_path / f"{prefix}{i}.wtns").read_bytes()).hexdigest() == c["wtns_sha256"], c["name"]
        for i in (40, 41, 127, 128, 167, 255, 256, 299):
            assert (tmp_path / f"{prefix}{i}.wtns").read_bytes() == hdr + want[i - 40].tobytes(), (threads, i)
        assert len([f for f in os.listdir(tmp_path) if f.startswith(prefix)]) == recs.shape[0]
        for f in os.listdir(tmp_path):
            os.unlink(tmp_path / f)
    # a sub-range, and a directory that does not exist (every writer fails: one error, no hang)
    assert b.write_wtns(tmp_path, "sub_", first=130, count=7, threads=3) == 7 and sorted(os.listdir(tmp_path)) == sorted(f"sub_{i}.wtns" for i in range(130, 137))
    with pytest.raises(m.B3WError) as ei:
        b.write_wtns(tmp_path / "missing", "x", threads=4)
    assert "cannot write" in str(ei.value)
    b.close(); ctx.close()
    # nova: a rejected step produces no file
    nrecs = W.config3_nova(9, first=5).copy()
    nrecs[4, 14] = nrecs[4, 12]
    ctx = _ctx(m, "nova_vesta")
    b = m.Batch(ctx, 9)
    b.run(nrecs)
    assert b.write_wtns(tmp_path, "n") == 8 and not (tmp_path / "n4.wtns").exists()
    _, want = T.oracle_batch_u32("nova_vesta", nrecs)
    assert (tmp_path / "n8.wtns").read_bytes() == T.oracle_header("nova_vesta") + want[8].tobytes()
    b.close(); ctx.close()


def test_batch_launch_inside_a_hip_graph(m):
    """b3w_batch_run_device only launches (no allocation, no synchronisation): it can be captured into a hipGraph and
    replayed; the replay on NEW records must produce those records' witnesses."""
    import torch
    n = 64
    ctx = _ctx(m, "compression")
    dev = torch.device("cuda:0")
    recs_a = T.workloads().config2_compression(n, first=10)
    recs_b = T.workloads().config2_compression(n, first=5000)
    d_recs = torch.from_numpy(recs_a.view(np.int32)).to(dev)
    d_bodies = torch.zeros((n, ctx.body_bytes), dtype=torch.uint8, device=dev)
    d_pub = torch.zeros((n, 16), dtype=torch.int32, device=dev)
    d_st = torch.full((n,), -1, dtype=torch.int32, device=dev)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                                       # warm-up outside the capture
        ctx.run_device(d_recs.data_ptr(), n, d_bodies.data_ptr(), 0, d_pub.data_ptr(), d_st.data_ptr(), side.cuda_stream)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with T.pkg().graph_capture(g, stream=side):
        ctx.run_device(d_recs.data_ptr(), n, d_bodies.data_ptr(), 0, d_pub.data_ptr(), d_st.data_ptr(), torch.cuda.current_stream().cuda_stream)
    d_recs.copy_(torch.from_numpy(recs_b.view(np.int32)))                 # new inputs, same buffers
    d_bodies.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert int(d_st.abs().sum().item()) == 0
    _, want = T.oracle_batch_u32("compression", recs_b)
    got = d_bodies.cpu().numpy()
    for i in (0, 17, n - 1):
        assert np.array_equal(got[i], want[i]), i
    ctx.close()
