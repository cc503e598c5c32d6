"""The on-device rank-1 constraint check (b3w_r1cs_check_device, csrc/b3w_r1cs.hip) — the GPU counterpart of
circom_tester's expectPass / expectFail in the reference's tests (test/blake3_hash.test.ts:36,44).  The constraint
system is the one tools/gen_r1cs.py derives from the circuit text (pinned on the CPU by tests/test_r1cs_cpu.py against
the reference's own witness); the kernel's verdicts are compared with a plain-integer evaluation (tests/r1cs_ref.py).
None of this runs the witness kernels' TRACE code on the bodies under test."""
import ctypes
import os
import random

import numpy as np
import pytest

import b3w_testlib as T
import r1cs_ref as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    m = T.pkg()
    ctx = m.Context("compression", 0)
    return dict(m=m, ctx=ctx, r1cs=m.R1cs(ctx), sys=R.parse(R.read_image()), torch=torch, dev=torch.device("cuda:0"))


def _check(env, bodies):
    """bodies: uint8 [n, body_bytes] CUDA tensor -> (violations, first) as numpy"""
    torch = env["torch"]
    n = bodies.shape[0]
    viol = torch.full((n,), 77, dtype=torch.int32, device=env["dev"])
    first = torch.zeros((n,), dtype=torch.int32, device=env["dev"])
    env["r1cs"].check_device(bodies.data_ptr(), n, bodies.stride(0), viol.data_ptr(), first.data_ptr(),
                             torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return viol.cpu().numpy().view(np.uint32), first.cpu().numpy().view(np.uint32)


def test_the_loaded_system_is_the_files(env):
    r = env["r1cs"]
    assert (r.n_constraints, r.n_wires, r.n_terms) == (24544, 24093, 117760)
    assert (r.n_pub_out, r.n_pub_in, r.n_prv_in) == (16, 0, 28)


def test_reference_witness_passes(env):
    """build/blake3_compression/testInp/witness.wtns as committed by the reference: 0 violations."""
    torch = env["torch"]
    body = np.frombuffer(T.golden_image("reference_testInp_witness.wtns.gz")[76:], dtype=np.uint8).copy()
    viol, first = _check(env, torch.from_numpy(body).to(env["dev"]).view(1, -1))
    assert viol[0] == 0 and first[0] == 0xFFFFFFFF


def test_kernel_witnesses_pass_and_1000_single_slot_corruptions_fail(env):
    """A clean config-2 batch satisfies every constraint; then one slot of each of 1 000 bodies is changed (bit flips,
    off-by-ones, random elements, a word in a bit slot): every one is caught, and count and first violated constraint
    equal the plain-integer evaluation."""
    torch, m, ctx, sys_ = env["torch"], env["m"], env["ctx"], env["sys"]
    n = 1000
    recs = m.workloads.config2_compression(n, first=50000)
    d_recs = torch.from_numpy(recs.view(np.int32)).to(env["dev"])
    bodies = torch.empty((n, ctx.body_bytes), dtype=torch.uint8, device=env["dev"])
    st = torch.zeros(n, dtype=torch.int32, device=env["dev"])
    ctx.run_device(d_recs.data_ptr(), n, bodies.data_ptr(), 0, 0, st.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert int(st.abs().sum().item()) == 0
    viol, first = _check(env, bodies)
    assert (viol == 0).all() and (first == 0xFFFFFFFF).all()

    rng = random.Random(2024)
    p = sys_["prime"]
    by_wire = R.rows_of_wire(sys_)
    host = bodies.cpu().numpy()
    slots = rng.sample(range(24093), 900) + [0, 1, 16, 17, 44, 45, 24092] + [rng.randrange(24093) for _ in range(93)]
    changed = []
    for i, s in enumerate(slots):
        old = int.from_bytes(host[i, 32 * s:32 * s + 32].tobytes(), "little")
        new = rng.choice([old ^ 1, (old + 1) % p, (old - 1) % p, rng.randrange(p), old ^ (1 << rng.randrange(34)), 2 if old < 2 else 0])
        if new == old:
            new = (old + 5) % p
        host[i, 32 * s:32 * s + 32] = np.frombuffer(new.to_bytes(32, "little"), dtype=np.uint8)
        changed.append((s, new))
    viol, first = _check(env, torch.from_numpy(host).to(env["dev"]))
    assert (viol > 0).all(), np.nonzero(viol == 0)[0][:10]
    for i in range(0, n, 9):                          # every 9th body against the integer evaluation of the rows that read the slot
        s, new = changed[i]
        z = R.body_to_ints(host[i])
        want = R.violated(sys_, z, by_wire[s])
        assert viol[i] == len(want) and first[i] == min(want), (i, s)


def test_accepted_non_canonical_inputs_pass_and_elements_past_p_fail(env):
    """The circuit has no range check on m: the WASM accepts m[0] = -1, 2^32, 2^33; those witnesses (exact
    kernel) satisfy the constraints too.  An element >= p is not a witness value even when congruent to the right one."""
    torch, m = env["torch"], env["m"]
    g = T.golden("compression")
    wild = [c for c in g["cases"] if "error" not in c and not T.is_canonical_u32("compression", c["input"])][:6]
    assert wild
    wc = m.WitnessCalculator(env["ctx"])
    bodies = np.stack([wc.calculateBinWitness(c["input"], 0) for c in wild])
    viol, _ = _check(env, torch.from_numpy(bodies).to(env["dev"]))
    assert (viol == 0).all(), [c["name"] for c, v in zip(wild, viol) if v]
    p = env["sys"]["prime"]
    z = R.body_to_ints(bodies[0])
    s = 20000
    bodies[0, 32 * s:32 * s + 32] = np.frombuffer((z[s] + p).to_bytes(32, "little"), dtype=np.uint8)
    viol, first = _check(env, torch.from_numpy(bodies).to(env["dev"]))
    want = R.violated(env["sys"], R.body_to_ints(bodies[0]))
    assert viol[0] == len(want) > 0 and first[0] == min(want) and (viol[1:] == 0).all()


def test_padded_pitch_and_batch_entry_point(env):
    torch, m, ctx = env["torch"], env["m"], env["ctx"]
    recs = m.workloads.config2_compression(5, first=123)
    b = m.Batch(ctx, 8, pitch=ctx.body_bytes + 96)
    b.run(recs)
    viol, first = b.r1cs_check(env["r1cs"])
    assert (viol == 0).all() and (first == 0xFFFFFFFF).all() and len(viol) == 5
    b.close()


def test_bad_images_are_refused(env):
    m, ctx = env["m"], env["ctx"]
    img = R.read_image()
    for bad, why in ((img[:1000], "truncated"), (b"r1cx" + img[4:], "not an r1cs"), (img[:4] + b"\x02" + img[5:], "version")):
        with pytest.raises(m.B3WError) as e:
            m.R1cs(ctx, bad)
        assert why in str(e.value), str(e.value)
    nova = m.Context("nova_vesta", 0)
    with pytest.raises(m.B3WError) as e:
        m.R1cs(nova, img)
    assert "prime" in str(e.value)
    m.R1cs(nova).close()                               # its own derived system loads
    nova.close()
    nb = m.Context("nova_bn254", 0)
    with pytest.raises(m.B3WError) as e:
        m.R1cs(nb, img)
    assert "nWires" in str(e.value)
    nb.close()


def test_full_config2_batch_passes(env):
    """All 4 096 witnesses of BASELINE config 2 satisfy all 24 544 constraints (100 M constraint evaluations)."""
    torch, m, ctx = env["torch"], env["m"], env["ctx"]
    n = 4096
    recs = m.workloads.config2_compression(n)
    d_recs = torch.from_numpy(recs.view(np.int32)).to(env["dev"])
    bodies = torch.empty((n, ctx.body_bytes), dtype=torch.uint8, device=env["dev"])
    ctx.run_device(d_recs.data_ptr(), n, bodies.data_ptr(), 0, 0, 0, torch.cuda.current_stream().cuda_stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    viol = torch.zeros(n, dtype=torch.int32, device=env["dev"])
    env["r1cs"].check_device(bodies.data_ptr(), n, 0, viol.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
    e0.record()
    env["r1cs"].check_device(bodies.data_ptr(), n, 0, viol.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
    e1.record()
    torch.cuda.synchronize()
    assert int(viol.abs().sum().item()) == 0
    print(f"r1cs check: {n} bodies x 24544 constraints in {e0.elapsed_time(e1):.2f} ms")


def test_nova_o1_build_on_the_device():
    """The circomkit build of blake3_nova (BN254, 24 614 wires): its derived constraint system (wires by circom's numbering
    rule, pinned on the CPU against the reference WASM's witnesses) on the device — a clean config-3 batch incl. parent
    steps satisfies all 25 067 constraints; every single-slot change is caught, count and first violated row equal the
    integer evaluation; goldens of accepted non-canonical inputs (exact kernel) pass."""
    import torch
    m = T.pkg()
    dev = torch.device("cuda:0")
    ctx = m.Context("nova_bn254_o1", 0)
    r1cs = m.R1cs(ctx)
    assert (r1cs.n_constraints, r1cs.n_wires, r1cs.n_pub_out, r1cs.n_pub_in, r1cs.n_prv_in) == (25067, 24614, 15, 12, 20)
    sys_ = R.parse(R.read_image(R.BUILTIN_NOVA_O1))
    n = 600
    recs = m.workloads.config3_nova(n, first=777)
    d_recs = torch.from_numpy(recs.view(np.int32)).to(dev)
    bodies = torch.empty((n, ctx.body_bytes), dtype=torch.uint8, device=dev)
    st = torch.zeros(n, dtype=torch.int32, device=dev)
    ctx.run_device(d_recs.data_ptr(), n, bodies.data_ptr(), 0, 0, st.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert int(st.abs().sum().item()) == 0

    def check(b):
        viol = torch.full((b.shape[0],), 77, dtype=torch.int32, device=dev)
        first = torch.zeros((b.shape[0],), dtype=torch.int32, device=dev)
        r1cs.check_device(b.data_ptr(), b.shape[0], b.stride(0), viol.data_ptr(), first.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        return viol.cpu().numpy().view(np.uint32), first.cpu().numpy().view(np.uint32)
    viol, first = check(bodies)
    assert (viol == 0).all() and (first == 0xFFFFFFFF).all()
    rng = random.Random(5)
    p = sys_["prime"]
    by_wire = R.rows_of_wire(sys_)
    host = bodies.cpu().numpy()
    slots = rng.sample(range(24614), n - 8) + [0, 1, 15, 16, 28, 45, 46, 24613]
    for i, s in enumerate(slots):
        old = int.from_bytes(host[i, 32 * s:32 * s + 32].tobytes(), "little")
        new = rng.choice([old ^ 1, (old + 1) % p, (old - 1) % p, rng.randrange(p), 2 if old < 2 else 0])
        if new == old:
            new = (old + 5) % p
        host[i, 32 * s:32 * s + 32] = np.frombuffer(new.to_bytes(32, "little"), dtype=np.uint8)
    viol, first = check(torch.from_numpy(host).to(dev))
    assert (viol > 0).all(), np.nonzero(viol == 0)[0][:10]
    for i in range(0, n, 7):
        want = R.violated(sys_, R.body_to_ints(host[i]), by_wire[slots[i]])
        assert viol[i] == len(want) and first[i] == min(want), (i, slots[i])
    g = T.golden("nova_bn254_o1")
    wild = [c for c in g["cases"] if "error" not in c and not T.is_canonical_u32("nova_bn254_o1", c["input"])][:5]
    wc = m.WitnessCalculator(ctx)
    wb = np.stack([wc.calculateBinWitness(c["input"], 0) for c in wild])
    viol, _ = check(torch.from_numpy(wb).to(dev))
    assert (viol == 0).all(), [c["name"] for c, v in zip(wild, viol) if v]
    r1cs.close(); ctx.close()


@pytest.mark.parametrize("circuit", ["nova_vesta", "nova_bn254"])
def test_nova_o2_builds_on_the_device(circuit):
    """The builds the reference folds with (rust_fold/src/main.rs:29,364; BASELINE configs 3-5): derived systems over their
    23 291 wires (tools/gen_r1cs.py --circuit nova_o2, pinned on the CPU against the reference WASM's witnesses).  A clean
    config-3 batch satisfies all 23 744 constraints; single-slot changes are caught with the integer evaluation's count and
    first row; witnesses of accepted non-canonical inputs (exact kernel) pass."""
    import torch
    m = T.pkg()
    dev = torch.device("cuda:0")
    ctx = m.Context(circuit, 0)
    r1cs = m.R1cs(ctx)
    assert (r1cs.n_constraints, r1cs.n_wires, r1cs.n_pub_out, r1cs.n_pub_in, r1cs.n_prv_in) == (23744, 23291, 15, 12, 20)
    sys_ = R.parse(R.read_image(R.BUILTIN_NOVA_O2[circuit]))
    n = 600
    recs = m.workloads.config3_nova(n, first=4242)
    d_recs = torch.from_numpy(recs.view(np.int32)).to(dev)
    bodies = torch.empty((n, ctx.body_bytes), dtype=torch.uint8, device=dev)
    st = torch.zeros(n, dtype=torch.int32, device=dev)
    ctx.run_device(d_recs.data_ptr(), n, bodies.data_ptr(), 0, 0, st.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert int(st.abs().sum().item()) == 0

    def check(b):
        viol = torch.full((b.shape[0],), 77, dtype=torch.int32, device=dev)
        first = torch.zeros((b.shape[0],), dtype=torch.int32, device=dev)
        r1cs.check_device(b.data_ptr(), b.shape[0], b.stride(0), viol.data_ptr(), first.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        return viol.cpu().numpy().view(np.uint32), first.cpu().numpy().view(np.uint32)
    viol, first = check(bodies)
    assert (viol == 0).all() and (first == 0xFFFFFFFF).all()
    rng = random.Random(6)
    p = sys_["prime"]
    by_wire = R.rows_of_wire(sys_)
    host = bodies.cpu().numpy()
    slots = rng.sample(range(23291), n - 8) + [0, 1, 15, 16, 28, 45, 46, 23290]
    for i, s in enumerate(slots):
        old = int.from_bytes(host[i, 32 * s:32 * s + 32].tobytes(), "little")
        new = rng.choice([old ^ 1, (old + 1) % p, (old - 1) % p, rng.randrange(p), 2 if old < 2 else 0])
        if new == old:
            new = (old + 5) % p
        host[i, 32 * s:32 * s + 32] = np.frombuffer(new.to_bytes(32, "little"), dtype=np.uint8)
    viol, first = check(torch.from_numpy(host).to(dev))
    assert (viol > 0).all(), np.nonzero(viol == 0)[0][:10]
    for i in range(0, n, 7):
        want = R.violated(sys_, R.body_to_ints(host[i]), by_wire[slots[i]])
        assert viol[i] == len(want) and first[i] == min(want), (i, slots[i])
    g = T.golden(circuit)
    wild = [c for c in g["cases"] if "error" not in c and not T.is_canonical_u32(circuit, c["input"])][:5]
    wc = m.WitnessCalculator(ctx)
    wb = np.stack([wc.calculateBinWitness(c["input"], 0) for c in wild])
    viol, _ = check(torch.from_numpy(wb).to(dev))
    assert (viol == 0).all(), [c["name"] for c, v in zip(wild, viol) if v]
    r1cs.close(); ctx.close()


@pytest.mark.parametrize("circuit", ["nova_vesta", "nova_bn254_o1"])
def test_wide_elements_of_a_nova_step_tampered(circuit):
    """The rows the stream kernel defers in a VALID nova step are those over full field elements: the 67 IsZero gadgets'
    `in * inv = 1 - out` and `in * out = 0` (in = depth - i and the like, a small signed number; inv a 254-bit inverse).  The deferred
    kernel decides (small signed) x (field element) = c without a field multiplication (small_product_is, csrc/b3w_r1cs_device.h).  Every
    wide slot of a step witness is changed in several ways — neighbours, 0, 1, -1, another gadget's inverse, the edges of "small"
    (2^32 - 1, 2^32, p - 2^32 + 1, p - 2^32), random — and count and first violated row must equal the plain-integer evaluation."""
    import torch
    m = T.pkg()
    dev = torch.device("cuda:0")
    ctx = m.Context(circuit, 0)
    r1cs = m.R1cs(ctx)
    sys_ = R.parse(R.read_image(R.BUILTIN_NOVA_O2[circuit] if circuit in R.BUILTIN_NOVA_O2 else R.BUILTIN_NOVA_O1))
    p = sys_["prime"]
    by_wire = R.rows_of_wire(sys_)
    recs = m.workloads.config3_nova(3, first=77)
    d_recs = torch.from_numpy(recs.view(np.int32)).to(dev)
    clean = torch.empty((3, ctx.body_bytes), dtype=torch.uint8, device=dev)
    ctx.run_device(d_recs.data_ptr(), 3, clean.data_ptr(), 0, 0, 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    clean = clean.cpu().numpy()
    rng = random.Random(11)
    cases = []                                               # (body of origin, slot, new value)
    for b in range(3):
        z = R.body_to_ints(clean[b])
        wide = [w for w, v in enumerate(z) if v >= 1 << 64]
        assert len(wide) >= 60, len(wide)
        for w in wide:
            old = z[w]
            other = z[rng.choice(wide)]
            for new in {(old + 1) % p, (old - 1) % p, 0, 1, p - 1, other, (1 << 32) - 1, 1 << 32, p - (1 << 32) + 1, p - (1 << 32), rng.randrange(p)}:
                if new != old and rng.random() < 0.4:
                    cases.append((b, w, new))
    host = np.stack([clean[b] for b, _, _ in cases])
    for i, (_, w, new) in enumerate(cases):
        host[i, 32 * w:32 * w + 32] = np.frombuffer(new.to_bytes(32, "little"), dtype=np.uint8)
    d = torch.from_numpy(host).to(dev)
    viol = torch.zeros(len(cases), dtype=torch.int32, device=dev)
    first = torch.zeros(len(cases), dtype=torch.int32, device=dev)
    r1cs.check_device(d.data_ptr(), len(cases), 0, viol.data_ptr(), first.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    viol, first = viol.cpu().numpy().view(np.uint32), first.cpu().numpy().view(np.uint32)
    assert len(cases) > 500
    for i, (b, w, new) in enumerate(cases):
        want = R.violated(sys_, R.body_to_ints(host[i]), by_wire[w])
        assert viol[i] == len(want) and first[i] == (min(want) if want else 0xFFFFFFFF), (i, b, w, hex(new), int(viol[i]), want[:4])
    r1cs.close(); ctx.close()


def test_gather_kernel_gives_the_same_verdicts(tmp_path):
    """csrc/b3w_r1cs_walk.hip and b3w_r1cs.hip hold three formulations with the same verdicts: the walk kernel (default since round 4: a workgroup walks whole
    bodies, earlier tiles' wires come from an export area in LDS, truth-table rows in runs), the stream kernel (B3W_R1CS_GATHER=4:
    tile-major units, outside wires gathered; the fallback for tiled systems the walk kernel has no room for) and the gather kernel
    (any system; B3W_R1CS_GATHER=1).  (Round 2's lean pair, a fourth until round 4, is gone.)  Child processes run each over the same clean and
    corrupted bodies; counts and first violated rows must be identical.  (n = 300 bodies on up to 512 workgroups: one body each;
    B3W_R1CS_GRID=7: 42 - 43 bodies per workgroup, body after body through the pipeline.)"""
    import json, os, subprocess, sys
    script = r'''
import importlib, json, os, sys, random
import numpy as np, torch
sys.path.insert(0, os.getcwd())
m = importlib.import_module("hot-proofs-blake3-circom_amd")
out = {}
for circuit in ("compression", "nova_bn254_o1", "nova_vesta", "nova_bn254"):
    ctx = m.Context(circuit, 0)
    r1cs = m.R1cs(ctx)
    n = 300
    recs = m.workloads.config2_compression(n, first=9) if circuit == "compression" else m.workloads.config3_nova(n, first=9)
    d_recs = torch.from_numpy(recs.view(np.int32)).cuda()
    bodies = torch.empty((n, ctx.body_bytes), dtype=torch.uint8, device="cuda")
    ctx.run_device(d_recs.data_ptr(), n, bodies.data_ptr(), 0, 0, 0, torch.cuda.current_stream().cuda_stream)
    host = bodies.cpu().numpy()
    rng = random.Random(1)
    for i in range(0, n, 2):                      # every other body: one slot changed
        s = rng.randrange(ctx.witness_size)
        host[i, 32 * s:32 * s + 32] = np.frombuffer(rng.randrange(1 << 255).to_bytes(32, "little"), dtype=np.uint8)
    if circuit in ("nova_vesta", "nova_bn254"):   # ... and, O2 builds, the field inverses of the IsZero gadgets (slots 23100 + 2 j: the rows the walk
        for j, i in enumerate(range(0, 90, 2)):   # kernel hands over as wide records): off by one, a small number, no canonical representative
            s = 23100 + 2 * j
            v = int.from_bytes(host[i, 32 * s:32 * s + 32].tobytes(), "little")
            host[i, 32 * s:32 * s + 32] = np.frombuffer([v + 1, 5, (1 << 255) + 3][j % 3].to_bytes(32, "little"), dtype=np.uint8)
        for j, i in enumerate(range(90, 130, 2)):  # ... and one bit of Num2Bits(65)(chunk_idx) flipped: the always-deferred row of 133 terms (the deferred kernel's own list)
            host[i, 32 * (23230 + 3 * j)] ^= 1
    d = torch.from_numpy(host).cuda()
    viol = torch.zeros(n, dtype=torch.int32, device="cuda"); first = torch.zeros(n, dtype=torch.int32, device="cuda")
    r1cs.check_device(d.data_ptr(), n, 0, viol.data_ptr(), first.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    out[circuit] = [viol.cpu().numpy().view(np.uint32).tolist(), first.cpu().numpy().view(np.uint32).tolist()]
print(json.dumps(out))
'''
    res = {}
    for mode, extra in (("0", {}), ("1", {}), ("4", {}), ("4/16waves", {"B3W_R1CS_WAVES": "16"}), ("4/grid7", {"B3W_R1CS_GRID": "7"}),
                        ("0/grid7", {"B3W_R1CS_GRID": "7"}), ("0/grid1", {"B3W_R1CS_GRID": "1"}),
                        # the walk kernel's two instantiations, each for every circuit (by default the circomkit build takes the signed one — small
                        # negative numbers p - k count as -k — and the others the unsigned one)
                        ("0/signed", {"B3W_R1CS_SIGNED": "1"}), ("0/unsigned", {"B3W_R1CS_SIGNED": "0"})):
        r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, cwd=T.ROOT, timeout=600,
                           env=dict(os.environ, B3W_R1CS_GATHER=mode.split("/")[0], **extra))
        assert r.returncode == 0, r.stderr[-1500:]
        res[mode] = json.loads(r.stdout.strip().splitlines()[-1])
    for mode in res:
        for circuit in res["1"]:
            assert res[mode][circuit] == res["1"][circuit], (mode, circuit, [(i, a, b) for i, (a, b) in enumerate(zip(res[mode][circuit][0], res["1"][circuit][0])) if a != b][:5])
    for circuit in res["0"]:
        viol = res["0"][circuit][0]
        assert all(v == 0 for v in viol[1::2]) and all(v > 0 for v in viol[0::2]), circuit


def test_mutated_images_never_crash_the_parser(env):
    """The .r1cs image is untrusted input: 400 random mutations of the derived file (truncations, flipped header and count
    bytes, absurd constraint counts) are either refused with an error text or load as a (different) system — no exception
    crosses the C boundary, nothing is read out of bounds."""
    m, ctx = env["m"], env["ctx"]
    img = bytearray(R.read_image())
    rng = random.Random(99)
    refused = loaded = 0
    for k in range(400):
        bad = bytearray(img)
        kind = k % 4
        if kind == 0:
            bad = bad[:rng.randrange(0, len(bad))]
        elif kind == 1:                                    # the header section (after the 12-byte preamble and its 12-byte section head)
            for _ in range(rng.randrange(1, 4)):
                bad[rng.randrange(0, 100)] = rng.randrange(256)
        elif kind == 2:                                    # a term count or wire index somewhere in the constraints
            pos = rng.randrange(100, len(bad) - 4)
            bad[pos:pos + 4] = rng.randrange(1 << 32).to_bytes(4, "little")
        else:                                              # mConstraints: absurdly large
            bad[24 + 36 + 24:24 + 36 + 28] = rng.choice([0xFFFFFFFF, 0x7FFFFFFF, 1 << 30]).to_bytes(4, "little")
        try:
            r = m.R1cs(ctx, bytes(bad))
            r.close()
            loaded += 1
        except m.B3WError as e:
            assert e.status in (100, 5) and "r1cs" in str(e), str(e)
            refused += 1
    assert refused >= 230 and refused + loaded == 400


def test_slabs_and_streams(env):
    """The lean kernel pair checks in slabs of 8 192 bodies through one deferred-row scratch per stream: a batch that spans
    two slabs with corrupted bodies on both sides of the border is judged body by body like the same bodies checked alone,
    and two streams checking different batches with one R1cs object at the same time do not disturb each other."""
    torch, m, ctx = env["torch"], env["m"], env["ctx"]
    dev = env["dev"]
    n = 8192 + 300
    recs = m.workloads.config2_compression(n, first=123)
    d_recs = torch.from_numpy(recs.view(np.int32)).to(dev)
    bodies = torch.empty((n, ctx.body_bytes), dtype=torch.uint8, device=dev)
    ctx.run_device(d_recs.data_ptr(), n, bodies.data_ptr(), 0, 0, 0, torch.cuda.current_stream().cuda_stream)
    rng = random.Random(5)
    hit = sorted(rng.sample(range(8000, n), 60) + [0, 8191, 8192, n - 1])
    elems = bodies.view(n, ctx.witness_size, 32)
    for i in hit:                                            # a random element (almost surely >= 2^63: a deferred row) or a 2
        s = rng.randrange(1, ctx.witness_size)
        val = rng.randrange(1 << 250) if i % 2 else 2
        elems[i, s] = torch.from_numpy(np.frombuffer(val.to_bytes(32, "little"), dtype=np.uint8).copy()).to(dev)
    viol, first = _check(env, bodies)
    got = set(np.nonzero(viol)[0].tolist())
    assert got == set(hit), (sorted(got - set(hit))[:10], sorted(set(hit) - got)[:10])
    alone_v, alone_f = _check(env, bodies[hit].contiguous())
    assert np.array_equal(viol[hit], alone_v) and np.array_equal(first[hit], alone_f)
    # two streams, one R1cs object
    r1cs = env["r1cs"]
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    a, b = bodies[:3000], bodies[8000:8400].contiguous()
    va = torch.full((a.shape[0],), 7, dtype=torch.int32, device=dev)
    vb = torch.full((b.shape[0],), 7, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    for _ in range(4):
        r1cs.check_device(a.data_ptr(), a.shape[0], 0, va.data_ptr(), 0, sa.cuda_stream)
        r1cs.check_device(b.data_ptr(), b.shape[0], 0, vb.data_ptr(), 0, sb.cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(va.cpu().numpy().view(np.uint32), viol[:3000])
    assert np.array_equal(vb.cpu().numpy().view(np.uint32), viol[8000:8400])


def test_three_slabs_of_nova_steps():
    """A check of more than 8 192 bodies goes slab by slab through one scratch (blocks, body words, wide records).  3 x 8 192 + 100
    nova steps — every one with 67 wide records and the always-deferred row — with tampered inverses, tampered bits and random
    elements on both sides of every slab border: verdicts and first rows equal those of the same bodies checked alone, twice in a row."""
    import torch
    m = T.pkg()
    dev = torch.device("cuda:0")
    ctx = m.Context("nova_vesta", 0)
    r1cs = m.R1cs(ctx)
    n = 3 * 8192 + 100
    recs = m.workloads.config3_nova(n, first=5)
    d_recs = torch.from_numpy(recs.view(np.int32)).to(dev)
    bodies = torch.empty((n, ctx.body_bytes), dtype=torch.uint8, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    ctx.run_device(d_recs.data_ptr(), n, bodies.data_ptr(), 0, 0, 0, s)
    torch.cuda.synchronize()
    z0 = R.body_to_ints(bodies[0].cpu().numpy())
    wide = [w for w, v in enumerate(z0) if v >= 1 << 64]
    rng = random.Random(21)
    hit = sorted(set([0, 8191, 8192, 16383, 16384, 24575, 24576, n - 1] + rng.sample(range(n), 150)))
    elems = bodies.view(n, ctx.witness_size, 32)
    for k, i in enumerate(hit):
        if k % 3 == 0:                                       # an inverse of its own step, off by one: a wide record says no
            w = rng.choice(wide)
            old = int.from_bytes(bytes(elems[i, w].cpu().numpy()), "little")
            val = old + 1 if old else 5
        elif k % 3 == 1:                                     # a 2 where a bit belongs (or a general row's operand)
            w, val = rng.randrange(1, ctx.witness_size), 2
        else:                                                # a random 250-bit element
            w, val = rng.randrange(1, ctx.witness_size), rng.randrange(1 << 250)
        elems[i, w] = torch.from_numpy(np.frombuffer(val.to_bytes(32, "little"), dtype=np.uint8).copy()).to(dev)

    def check(b):
        k = b.shape[0]
        viol = torch.full((k,), 77, dtype=torch.int32, device=dev)
        first = torch.zeros((k,), dtype=torch.int32, device=dev)
        r1cs.check_device(b.data_ptr(), k, b.stride(0), viol.data_ptr(), first.data_ptr(), s)
        torch.cuda.synchronize()
        return viol.cpu().numpy().view(np.uint32), first.cpu().numpy().view(np.uint32)
    alone_v, alone_f = check(bodies[hit].contiguous())
    assert np.count_nonzero(alone_v) >= len(hit) - 3, np.count_nonzero(alone_v)       # (a random slot may be one no constraint reads)
    for _ in range(2):
        viol, first = check(bodies)
        bad = set(np.nonzero(viol)[0].tolist())
        assert bad <= set(hit), sorted(bad - set(hit))[:10]
        assert np.array_equal(viol[hit], alone_v) and np.array_equal(first[hit], alone_f)
    r1cs.close(); ctx.close()


@pytest.mark.parametrize("shape", ["local", "scattered"])
def test_random_systems_against_plain_integers(shape, tmp_path):
    """b3w_r1cs_create takes ANY iden3 .r1cs over the context's field.  Synthetic systems over the 24 093 wires — random
    field-element and small coefficients, rows of 0 ... 300 terms, empty parts, rows far from their wires ("scattered": more
    than 1 024 outside wires per tile, which only the gather kernel takes) — against random bodies (field elements, words,
    bits, elements >= p): counts and first violated rows of all three formulations equal the plain-integer evaluation."""
    import json, os, subprocess, sys
    p = R.parse(R.read_image())["prime"]
    nw = T.NWIT["compression"]
    rng = random.Random(7 if shape == "local" else 8)
    cons = []
    def lc(n, centre):
        out = {}
        for _ in range(n):
            w = rng.randrange(nw) if shape == "scattered" or rng.random() < 0.02 else min(nw - 1, max(0, centre + rng.randrange(-300, 300)))
            kind = rng.random()
            out[w] = (rng.randrange(p) if kind < 0.15 else p - rng.randrange(1, 1 << 20) if kind < 0.3 else
                      1 << rng.randrange(0, 62) if kind < 0.6 else rng.choice([1, p - 1, 2, 3]))
        return out
    for k in range(3000):
        centre = rng.randrange(nw)
        sizes = [rng.choice([0, 1, 1, 2, 3, 5, 34, 67]) for _ in range(3)]
        if k % 500 == 0:
            sizes[2] = 300
        cons.append((lc(sizes[0], centre), lc(sizes[1], centre), lc(sizes[2], centre)))
    # bit runs (the lean kernel folds 4 ... 64 terms  +-2^(k+i) * z[s+i]  over consecutive wires into one step): rows
    # run * 1 = w_x over wires that hold bits in every body, every length, shift, sign, across 64-element groups and the tile border
    runs = []
    if shape == "local":
        for j in range(60):
            ln = rng.choice([4, 5, 31, 32, 33, 63, 64, 64])
            s0 = rng.choice([5000, 5056 - ln // 2, 5120 - 1, 6144 - ln // 2, 5200 + j])       # (6144 = a tile border)
            k0 = rng.randrange(0, 62 - ln) if ln < 62 else 0
            sgn = rng.choice([1, -1])
            a = {s0 + i: (sgn * (1 << (k0 + i))) % p for i in range(ln)}
            if j % 3 == 0:
                a[7000 + j] = 5                               # a run inside a longer part
            runs.append((len(cons), a, 8000 + j))
            cons.append((a, {0: 1}, {8000 + j: 1}))
    img = R.write_image(p, nw, cons, 16, 0, 28)
    path = tmp_path / "random.r1cs"
    path.write_bytes(img)
    n = 24
    bodies = np.zeros((n, nw, 32), dtype=np.uint8)
    zs = []
    for i in range(n):
        z = []
        for w in range(nw):
            kind = rng.random()
            v = (1 if w == 0 and i % 5 else rng.randrange(2) if kind < 0.6 else rng.randrange(1 << 32) if kind < 0.8 else
                 rng.randrange(1 << 40) if kind < 0.85 else rng.randrange(p) if kind < 0.995 else p + rng.randrange(1 << 200))
            z.append(v)
        if shape == "local":
            for w in range(4990, 6300):
                z[w] = rng.randrange(2) if i != 3 else rng.randrange(3)          # (body 3: some of the run's elements are no bits)
            for row, a, wx in runs:
                z[wx] = (sum(cf * z[w] for w, cf in a.items()) * z[0] + (i & 1 if row % 2 else 0)) % p
        zs.append(z)
        bodies[i] = np.frombuffer(b"".join(v.to_bytes(32, "little") for v in z), dtype=np.uint8).reshape(nw, 32)
    np.save(tmp_path / "bodies.npy", bodies)
    want = []
    for z in zs:                                             # plain integers: an element >= p spoils every row that reads it
        bad = []
        for k, (a, b, c) in enumerate(cons):
            wild = any(z[w] >= p for part in (a, b, c) for w in part)
            ev = lambda part: sum(cf * z[w] for w, cf in part.items())
            if wild or (ev(a) * ev(b) - ev(c)) % p:
                bad.append(k)
        want.append([len(bad), min(bad) if bad else 0xFFFFFFFF])
    script = r'''
import importlib, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
m = importlib.import_module("hot-proofs-blake3-circom_amd")
ctx = m.Context("compression", 0)
r = m.R1cs(ctx, open(sys.argv[1], "rb").read())
b = torch.from_numpy(np.load(sys.argv[2]).reshape(-1, ctx.body_bytes)).cuda()
n = b.shape[0]
viol = torch.zeros(n, dtype=torch.int32, device="cuda"); first = torch.zeros(n, dtype=torch.int32, device="cuda")
r.check_device(b.data_ptr(), n, 0, viol.data_ptr(), first.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
print(json.dumps([r.tiled] + [[int(x), int(y)] for x, y in zip(viol.cpu().numpy().view(np.uint32), first.cpu().numpy().view(np.uint32))]))
'''
    for mode in ("0", "1", "4"):
        res = subprocess.run([sys.executable, "-c", script, str(path), str(tmp_path / "bodies.npy")], capture_output=True, text=True,
                             cwd=T.ROOT, timeout=600, env=dict(os.environ, B3W_R1CS_GATHER=mode))
        assert res.returncode == 0, res.stderr[-1500:]
        got = json.loads(res.stdout.strip().splitlines()[-1])
        assert got[0] == (shape == "local"), "the local system must take the tile kernels, the scattered one the gather kernel"
        got = got[1:]
        assert got == want, (shape, mode, [(i, g, w) for i, (g, w) in enumerate(zip(got, want)) if g != w][:5])


def test_experiment_switches_cannot_neuter_a_product_build():
    """B3W_R1CS_DBG / B3W_R1CS_STAMPS are compiled only into a diagnostic build (-DB3W_R1CS_DIAG).  With both set in the
    environment of a fresh process the product library still finds a tampered witness (and, were it a diagnostic build, a launch
    with phases switched off marks every body as violating instead of passing it)."""
    import subprocess, sys
    script = r"""
import numpy as np, torch, sys
sys.path.insert(0, %r)
import b3w_testlib as T
m = T.pkg()
ctx = m.Context("compression", 0)
n = 64
recs = m.workloads.config2_compression(n, first=5)
dev = torch.device("cuda", 0)
d_recs = torch.from_numpy(recs.view(np.int32)).to(dev)
bodies = torch.zeros((n, ctx.body_bytes), dtype=torch.uint8, device=dev)
ctx.run_device(d_recs.data_ptr(), n, bodies.data_ptr(), 0, 0, 0, 0)
bodies[7, 32 * 5000] ^= 1                      # one bit of one slot of body 7
r = m.R1cs(ctx)
viol = torch.full((n,), 12345, dtype=torch.int32, device=dev)
r.check_device(bodies.data_ptr(), n, ctx.body_bytes, viol.data_ptr(), 0, 0)
torch.cuda.synchronize()
v = viol.cpu().numpy().view(np.uint32)
print("VIOL", int(v[7]), int(np.count_nonzero(np.delete(v, 7))))
""" % os.path.join(T.ROOT, "tests")
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=300, cwd=T.ROOT,
                       env=dict(os.environ, B3W_R1CS_DBG="1", B3W_R1CS_STAMPS="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    line = [x for x in r.stdout.splitlines() if x.startswith("VIOL")][0].split()
    tampered, others = int(line[1]), int(line[2])
    assert tampered > 0, "a tampered witness passed the constraint check with B3W_R1CS_DBG set"
    assert others in (0, 63)                   # product build: only body 7; diagnostic build: every body is marked
