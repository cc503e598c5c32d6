#!/usr/bin/env python3
"""Generate tests/golden/*.json by running the reference's committed WASMs (build container only;
needs node and /root/reference).  The fixtures hold inputs and expected outputs only: sha256 of
the witness body / .wtns image, the first 16 slots, expected error text, and a few gz bodies.

  python tools/gen_golden.py
"""
import gzip, hashlib, importlib.util, json, os, sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import recover_layout as RL
import b3w_model as M

spec = importlib.util.spec_from_file_location("b3w_workloads", os.path.join(REPO, "hot-proofs-blake3-circom_amd", "workloads.py"))
W = importlib.util.module_from_spec(spec)
spec.loader.exec_module(W)

GOLD = os.path.join(REPO, "tests", "golden")
REF = RL.REF


def wtns_header(prime, nwit):
    import struct
    return (b"wtns" + struct.pack("<IIIQI", 2, 2, 1, 40, 32) + prime.to_bytes(32, "little") +
            struct.pack("<IIQ", nwit, 2, 32 * nwit))


def compression_cases():
    cases = []
    c1 = W.config1_cases()
    cases.append(("config1_testInp", W.record_to_input(c1[0], W.COMPRESSION_KEYS)))
    for i in range(1, 6):
        cases.append((f"hash_test_followon_{i}", W.record_to_input(c1[i], W.COMPRESSION_KEYS)))
    c2 = W.config2_compression(32)
    for i in range(32):
        cases.append((f"config2_{i}", W.record_to_input(c2[i], W.COMPRESSION_KEYS)))
    z = np.zeros(28, dtype=np.uint32)
    cases.append(("all_zero", W.record_to_input(z, W.COMPRESSION_KEYS)))
    cases.append(("all_ones", W.record_to_input(z + np.uint32(0xFFFFFFFF), W.COMPRESSION_KEYS)))
    base = W.record_to_input(c2[0], W.COMPRESSION_KEYS)

    def mod(f):
        o = json.loads(json.dumps(base))
        f(o)
        return o
    # non-canonical inputs the WASM accepts (SURVEY 8(b) domain facts) and the ones it rejects
    cases.append(("m0_2p32", mod(lambda o: o["m"].__setitem__(0, str(2**32)))))
    cases.append(("m0_2p33", mod(lambda o: o["m"].__setitem__(0, str(2**33)))))
    cases.append(("m0_neg1", mod(lambda o: o["m"].__setitem__(0, "-1"))))
    cases.append(("m7_neg5", mod(lambda o: o["m"].__setitem__(7, "-5"))))
    cases.append(("m_all_neg", mod(lambda o: o.__setitem__("m", [str(-(i + 1)) for i in range(16)]))))
    cases.append(("err_m0_2p34", mod(lambda o: o["m"].__setitem__(0, str(2**34)))))
    cases.append(("err_h0_2p32", mod(lambda o: o["h"].__setitem__(0, str(2**32)))))
    cases.append(("err_h5_2p32", mod(lambda o: o["h"].__setitem__(5, str(2**32)))))
    cases.append(("err_d_2p32", mod(lambda o: o.__setitem__("d", str(2**32)))))
    cases.append(("err_b_neg1", mod(lambda o: o.__setitem__("b", "-1"))))
    cases.append(("err_t1_2p32", mod(lambda o: o["t"].__setitem__(1, str(2**32)))))
    return cases


def nova_cases():
    cases = []
    c3 = W.config3_nova(40)
    for i in range(40):
        cases.append((f"config3_{i}", W.record_to_input(c3[i], W.NOVA_KEYS)))
    import random
    rng = random.Random(77)
    for i in range(0, 64, 5):
        for bit in (0, 1):
            cases.append((f"directed_i{i}_b{bit}", RL.nova_probe(rng, directed=(i, bit))))
    for i in range(8):
        cases.append((f"probe_{i}", RL.nova_probe(rng)))
    leaf_i = next(i for i in range(40) if c3[i][14] == c3[i][12] - 1)
    par_i = next(i for i in range(40) if c3[i][14] < c3[i][12] - 1)
    base = W.record_to_input(c3[leaf_i], W.NOVA_KEYS)
    pbase = W.record_to_input(c3[par_i], W.NOVA_KEYS)

    def mod(_b=None, **kw):
        o = json.loads(json.dumps(_b or base))
        o.update(kw)
        return o
    # [0u8;4] single-block input (rust_fold/src/main.rs:478-485 shape): h=IV, one block, b=4
    single = dict(n_blocks=1, block_count=0, h=[int(x) for x in W.IV], chunk_idx_low=0, chunk_idx_high=0,
                  leaf_depth=1, total_depth=1, depth=0, m=[0] * 16, b=4)
    cases.append(("single_block_zero4", single))
    cases.append(("depth300_leaf301", mod(depth=300, leaf_depth=301)))
    cases.append(("total_depth_1000", mod(total_depth=1000)))
    cases.append(("n_blocks_0", mod(n_blocks=0)))
    cases.append(("b_max", mod(b=2**32 - 1)))
    cases.append(("block_count_big", mod(block_count=4000000000)))
    cases.append(("depth_big", mod(depth=3000000000, leaf_depth=3000000001, total_depth=3000000005)))
    cases.append(("m0_neg1", mod(m=["-1"] + base["m"][1:])))
    cases.append(("err_depth_ge_leaf", mod(depth=5, leaf_depth=5)))
    cases.append(("err_depth0_leaf0", mod(depth=0, leaf_depth=0)))
    cases.append(("err_b_2p32", mod(b=str(2**32))))
    cases.append(("err_cil_2p32", mod(chunk_idx_low=str(2**32))))
    cases.append(("err_h0_neg1", mod(h=["-1"] + base["h"][1:])))
    cases.append(("err_leaf_far", mod(depth=1, leaf_depth=600)))
    # parent steps mask h, chunk_idx and m[8..15] away, so these non-canonical values are accepted
    cases.append(("parent_cil_2p32", mod(pbase, chunk_idx_low=str(2**32))))
    cases.append(("parent_h0_neg1", mod(pbase, h=["-1"] + pbase["h"][1:])))
    cases.append(("parent_m9_neg1", mod(pbase, m=pbase["m"][:9] + ["-1"] + pbase["m"][10:])))
    return cases


def splitter_candidates(kind):
    """More inputs of the same families, for pin_slots below: the config-2 / config-3 streams behind the cases above, and seeded probes."""
    import random
    rng = random.Random(78)
    if kind == "compression":
        c2 = W.config2_compression(32 + 300)
        cand = [(f"config2_{i}", W.record_to_input(c2[i], W.COMPRESSION_KEYS)) for i in range(32, 332)]
        cand += [(f"probe_{i}", RL.compression_probe(rng)) for i in range(100)]
    else:
        c3 = W.config3_nova(40 + 300)
        cand = [(f"config3_{i}", W.record_to_input(c3[i], W.NOVA_KEYS)) for i in range(40, 340)]
        cand += [(f"probe_{i + 8}", RL.nova_probe(rng)) for i in range(200)]
        cand += [(f"directed_i{i}_b{bit}", RL.nova_probe(rng, directed=(i, bit))) for i in range(64) for bit in (0, 1)]      # (every eqs[i])
        cand += [(f"probe_{i + 208}", RL.nova_probe(rng)) for i in range(300)]
    return cand


def pin_slots(cfg, cases, bodies, errors):
    """Two witness slots that hold equal values in EVERY case above are not told apart by the fixtures: a layout that swapped them
    would pass.  Most such pairs are the same signal twice (a gadget's output bits and the next gadget's input bits) and stay
    equal whatever the input; the others are split here, by the reference's own witnesses of more inputs: a candidate is added as a
    case when it separates two slots no earlier case separates.  (tests/test_golden_pinning_cpu.py then checks that a few hundred
    further inputs of the CPU restatement separate nothing more.)"""
    ok = [i for i in range(len(cases)) if i not in errors]
    nwit = cfg["nwit"]

    def slot_keys(body):                                       # one 64-bit key per slot value
        w = np.frombuffer(body.tobytes(), dtype="<u8").reshape(nwit, 4)
        return w[:, 0] * np.uint64(0x9E3779B97F4A7C15) ^ w[:, 1] * np.uint64(0xC2B2AE3D27D4EB4F) ^ w[:, 2] * np.uint64(0x165667B19E3779F9) ^ w[:, 3] * np.uint64(0x27D4EB2F165667C5)

    def refine(gid, body):
        pair = np.stack([gid.astype(np.uint64), slot_keys(body)], axis=1)
        _, new = np.unique(pair, axis=0, return_inverse=True)
        return new.reshape(-1)

    gid = np.zeros(nwit, dtype=np.int64)
    for i in ok:
        gid = refine(gid, bodies[i])
    cand = splitter_candidates(cfg["kind"])
    cb, ce = RL.run_oracle(cfg["wasm"], [c[1] for c in cand])
    added = []
    for j, (name, inp) in enumerate(cand):
        if j in ce:
            continue
        new = refine(gid, cb[j])
        if new.max() > gid.max():
            added.append(("pin_" + name, inp, cb[j]))
            gid = new
    return added, int(gid.max()) + 1


def main():
    os.makedirs(GOLD, exist_ok=True)
    for circuit, cfg in RL.CIRCUITS.items():
        cases = compression_cases() if cfg["kind"] == "compression" else nova_cases()
        inputs = [c[1] for c in cases]
        bodies, errors = RL.run_oracle(cfg["wasm"], inputs)
        added, ngroups = pin_slots(cfg, cases, bodies, errors)
        print(f"{circuit}: {len(added)} pinning cases added; {ngroups} distinguishable slot groups of {cfg['nwit']} slots")
        if added:
            cases += [(a[0], a[1]) for a in added]
            bodies = np.concatenate([bodies, np.stack([a[2] for a in added])])
        hdr = wtns_header(cfg["prime"], cfg["nwit"])
        out = dict(circuit=circuit, prime=str(cfg["prime"]), nwit=cfg["nwit"],
                   generated_by="tools/gen_golden.py from the reference's committed " + cfg["wasm"],
                   cases=[])
        for i, (name, inp) in enumerate(cases):
            e = dict(name=name, input=inp)
            if i in errors:
                e["error"] = errors[i]
            else:
                body = bodies[i].tobytes()
                e["body_sha256"] = hashlib.sha256(body).hexdigest()
                e["wtns_sha256"] = hashlib.sha256(hdr + body).hexdigest()
                if not name.startswith("pin_"):
                    e["first16"] = [str(int.from_bytes(body[32 * s:32 * s + 32], "little")) for s in range(16)]
            out["cases"].append(e)
        with open(os.path.join(GOLD, f"{circuit}.json"), "w") as f:
            json.dump(out, f, indent=1)
        # a few full images, gz (about 10-12 KB each)
        keep = [0, 6] if cfg["kind"] == "compression" else [0, 3]
        for i in keep:
            with gzip.GzipFile(os.path.join(GOLD, f"{circuit}.{cases[i][0]}.wtns.gz"), "wb", mtime=0) as f:
                f.write(hdr + bodies[i].tobytes())
        nerr = len(errors)
        print(f"{circuit}: {len(cases)} cases, {nerr} expected-error cases")
        for i in sorted(errors):
            print("   ", cases[i][0], "->", json.dumps(errors[i])[:150])
    # the reference's own committed golden witness (data file held by the reference's tests)
    ref_w = open(os.path.join(REF, "build/blake3_compression/testInp/witness.wtns"), "rb").read()
    assert hashlib.sha256(ref_w).hexdigest() == "0c3f9a398e0683fd7d970429f2c2f2479a8cc246e2862bcd2afe7b17b783606f"
    with gzip.GzipFile(os.path.join(GOLD, "reference_testInp_witness.wtns.gz"), "wb", mtime=0) as f:
        f.write(ref_w)
    pub = json.load(open(os.path.join(REF, "build/blake3_compression/testInp/public.json")))
    json.dump(pub, open(os.path.join(GOLD, "reference_testInp_public.json"), "w"))


if __name__ == "__main__":
    main()
