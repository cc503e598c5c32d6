#!/usr/bin/env python3
"""Recover the witness slot layout of a committed circom WASM by value fingerprinting
(SURVEY.md Appendix B.2).  Build-container tool only: needs node + /root/reference.

  python tools/recover_layout.py <circuit> [--probes K] [--holdout H]

circuit in {compression, nova_bn254, nova_vesta, nova_bn254_o1}.  Writes
hot-proofs-blake3-circom_amd/layouts/<circuit>.layout:

  # header lines
  W <slot> <atom> <len>          slots slot..slot+len-1 hold atoms atom..atom+len-1 (whole value)
  B <slot> <atom> <bit0> <len>   slots hold bits bit0..bit0+len-1 of <atom>

Atom ids: tools/b3w_model.py.  The table is value-exact: where several names carry the same
value on every input (aliases / provable identities) any of them may be named.
"""
import argparse, json, os, subprocess, sys, tempfile, random
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import b3w_model as M

REF = os.environ.get("B3W_REFERENCE_DIR", "/root/reference")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAYOUT_DIR = os.path.join(REPO, "hot-proofs-blake3-circom_amd", "layouts")

CIRCUITS = {
    "compression": dict(wasm="build/blake3_compression/blake3_compression_js/blake3_compression.wasm",
                        prime=M.BN254_R, kind="compression", nwit=24093),
    "nova_bn254": dict(wasm="build/blake3_nova_js/blake3_nova.wasm", prime=M.BN254_R, kind="nova", nwit=23291),
    "nova_vesta": dict(wasm="build/blake3_nova_pasta_js/blake3_nova_pasta.wasm", prime=M.VESTA_Q, kind="nova", nwit=23291),
    "nova_bn254_o1": dict(wasm="build/blake3_nova/blake3_nova_js/blake3_nova.wasm", prime=M.BN254_R, kind="nova", nwit=24614),
}


def run_oracle(wasm_rel, inputs, nproc=8):
    """Returns (bodies uint8 [n, nwit*32], errors dict) from the reference WASM."""
    tmp = tempfile.mkdtemp(prefix="b3w_oracle_")
    inp = os.path.join(tmp, "in.json")
    def strs(x):
        if isinstance(x, dict):
            return {k: strs(v) for k, v in x.items()}
        if isinstance(x, (list, tuple)):
            return [strs(v) for v in x]
        return str(x)
    with open(inp, "w") as f:
        json.dump(strs(inputs), f)          # decimal strings: JSON numbers lose precision past 2^53
    n = len(inputs)
    per = (n + nproc - 1) // nproc
    procs = []
    for i in range(nproc):
        s = i * per
        if s >= n:
            break
        out = os.path.join(tmp, f"out{i}.bin")
        procs.append((out, subprocess.Popen(["node", os.path.join(REPO, "tools/wasm_oracle.js"),
                                             os.path.join(REF, wasm_rel), inp, out, str(s), str(per)])))
    bodies, errors = [], {}
    for out, pr in procs:
        if pr.wait() != 0:
            raise RuntimeError("oracle process failed")
        meta = json.load(open(out + ".err.json"))
        errors.update({int(k): v for k, v in meta["errors"].items()})
        bodies.append(np.fromfile(out, dtype=np.uint8).reshape(-1, meta["witnessSize"] * 32))
    subprocess.run(["rm", "-rf", tmp])
    return np.concatenate(bodies), errors


# ---------------------------------------------------------------- probe generators
def _wild_word(rng):
    """a message word outside [0,2^32) that the circuits still accept (SURVEY 8(b) domain facts)"""
    return rng.choice([-rng.randint(1, 2000), (1 << 32) + rng.getrandbits(32), rng.getrandbits(32)])


def compression_probe(rng, wild=False):
    m = [rng.getrandbits(32) for _ in range(16)]
    if wild:
        m = [_wild_word(rng) if rng.random() < 0.4 else x for x in m]
    return dict(h=[rng.getrandbits(32) for _ in range(8)], m=m,
                t=[rng.getrandbits(32), rng.getrandbits(32)], b=rng.getrandbits(32), d=rng.getrandbits(32))


def nova_probe(rng, directed=None, wild=False):
    pr = _nova_probe(rng, directed)
    if not wild:
        return pr
    big = lambda: rng.getrandbits(250)
    parent = pr["depth"] < pr["leaf_depth"] - 1
    if rng.random() < 0.5:
        pr["n_blocks"] = big()
    if rng.random() < 0.5:
        pr["block_count"] = big() if rng.random() < 0.7 else pr["n_blocks"] - 1
    if rng.random() < 0.3 and directed is None:
        pr["total_depth"] = big()
    if rng.random() < 0.3:
        x = big()
        delta = pr["leaf_depth"] - pr["depth"]
        if directed is not None:
            pr["total_depth"] = pr["total_depth"] - pr["depth"] + x
        pr["depth"], pr["leaf_depth"] = x, x + delta
    if parent:
        pr["m"] = pr["m"][:8] + [big() for _ in range(8)]
        pr["h"] = [_wild_word(rng) if rng.random() < 0.3 else v for v in pr["h"]]
        pr["m"] = [_wild_word(rng) if (i < 8 and rng.random() < 0.3) else v for i, v in enumerate(pr["m"])]
        if directed is None:
            pr["chunk_idx_low"] = rng.getrandbits(64)
    else:
        pr["m"] = [_wild_word(rng) if rng.random() < 0.3 else v for v in pr["m"]]
    return pr


def _nova_probe(rng, directed=None):
    n_blocks = rng.randint(1, 16)
    bc_kind = rng.randint(0, 3)
    block_count = 0 if bc_kind == 0 else (n_blocks - 1 if bc_kind == 1 else rng.randint(0, 40))
    depth = rng.choice([0, 0, 1, 2, 3, rng.randint(0, 60), rng.randint(0, 300)])
    parent = rng.random() < 0.5
    leaf_depth = depth + 1 if not parent else depth + 2 + rng.choice([0, 0, 1, 5, rng.randint(0, 200)])
    total_depth = rng.choice([leaf_depth, depth + 2 + rng.randint(0, 63), rng.randint(0, 90), 1000])
    cil, cih = rng.getrandbits(32), rng.getrandbits(32)
    if directed is not None:
        i, bit = directed
        total_depth = depth + i + 2
        ci = (cil | (cih << 32)) & ~(1 << i) | (bit << i)
        cil, cih = ci & 0xFFFFFFFF, ci >> 32
    return dict(n_blocks=n_blocks, block_count=block_count, h=[rng.getrandbits(32) for _ in range(8)],
                chunk_idx_low=cil, chunk_idx_high=cih, leaf_depth=leaf_depth, total_depth=total_depth,
                depth=depth, m=[rng.getrandbits(32) for _ in range(16)], b=rng.getrandbits(32))


def make_probes(kind, k, seed):
    rng = random.Random(seed)
    if kind == "compression":
        return [compression_probe(rng, wild=(j % 2 == 1)) for j in range(k)]
    probes = [nova_probe(rng, wild=(j % 2 == 1)) for j in range(k)]
    for i in range(64):
        for bit in (0, 1):
            for w in (False, True):
                probes.append(nova_probe(rng, directed=(i, bit), wild=w))
    return probes


def eval_atoms(kind, prime, inp):
    inp = {k: ([x % prime for x in v] if isinstance(v, list) else v % prime) for k, v in inp.items()}
    if kind == "compression":
        atoms = [0] * M.N_COMP_ATOMS
        M.eval_compression(prime, inp["h"], inp["m"], inp["t"], inp["b"], inp["d"], atoms)
    else:
        atoms = [0] * M.N_NOVA_ATOMS
        M.eval_nova(prime, inp, atoms)
    return atoms


def parse_layout(path):
    """-> list of (atom, bit) per slot; bit = -1 for whole."""
    slots = {}
    nwit = None
    for line in open(path):
        t = line.split()
        if not t or t[0].startswith("#"):
            if len(t) >= 3 and t[1] == "nwit":
                nwit = int(t[2])
            continue
        if t[0] == "W":
            s, a, ln = int(t[1]), int(t[2]), int(t[3])
            for j in range(ln):
                slots[s + j] = (a + j, -1)
        elif t[0] == "B":
            s, a, b0, ln = int(t[1]), int(t[2]), int(t[3]), int(t[4])
            for j in range(ln):
                slots[s + j] = (a, b0 + j)
    assert nwit is not None and len(slots) == nwit and all(i in slots for i in range(nwit))
    return [slots[i] for i in range(nwit)]


def predict_body(layout, atoms):
    out = bytearray(len(layout) * 32)
    for s, (a, bit) in enumerate(layout):
        v = atoms[a] if bit < 0 else (atoms[a] >> bit) & 1
        out[32 * s:32 * s + 32] = v.to_bytes(32, "little")
    return bytes(out)


def recover(circuit, k, holdout, seed=1234):
    cfg = CIRCUITS[circuit]
    kind, prime, nwit = cfg["kind"], cfg["prime"], cfg["nwit"]
    probes = make_probes(kind, k, seed)
    print(f"[{circuit}] {len(probes)} probes -> oracle", flush=True)
    bodies, errors = run_oracle(cfg["wasm"], probes)
    if errors:
        print(f"[{circuit}] dropping {len(errors)} probes the WASM rejected")
        keep = [q for q in range(len(probes)) if q not in errors]
        probes = [probes[q] for q in keep]
        bodies = bodies[keep]
    assert bodies.shape[1] == nwit * 32
    K = len(probes)
    atoms = [eval_atoms(kind, prime, p) for p in probes]          # [K][natoms]
    natoms = len(atoms[0])
    names = M.atom_names("compression" if kind == "compression" else "nova")
    # candidate fingerprints
    cand = {}
    def add(fp, ref):
        cand.setdefault(fp, []).append(ref)
    for a in range(natoms):
        col = [atoms[q][a] for q in range(K)]
        add(tuple(col), (a, -1))
        mx = max(col)
        if mx < (1 << 66):
            for bit in range(max(1, mx.bit_length())):
                add(tuple((c >> bit) & 1 for c in col), (a, bit))
            # bits above the observed maximum are all-zero columns: register up to the declared widths
            for bit in range(max(1, mx.bit_length()), 34 if mx < (1 << 34) else 65):
                add(tuple(0 for _ in col), (a, bit))
    # slot values
    slot_vals = []
    lim = bodies.reshape(K, nwit, 32)
    for s in range(nwit):
        slot_vals.append(tuple(int.from_bytes(lim[q, s].tobytes(), "little") for q in range(K)))
    layout = []
    unmatched = []
    prev = None
    for s in range(nwit):
        c = cand.get(slot_vals[s])
        if not c:
            unmatched.append(s)
            layout.append(None)
            prev = None
            continue
        pick = None
        if prev is not None:
            pa, pb = prev
            want = (pa, pb + 1) if pb >= 0 else (pa + 1, -1)
            if want in c:
                pick = want
        if pick is None:
            allbool = all(v in (0, 1) for v in slot_vals[s])
            # booleans: prefer a bit reference; words: whole reference. Lowest atom id first.
            pref = [r for r in c if (r[1] >= 0) == allbool] or c
            pick = min(pref, key=lambda r: (r[0], r[1]))
        layout.append(pick)
        prev = pick
    print(f"[{circuit}] unmatched slots: {len(unmatched)} {unmatched[:20]}")
    if unmatched:
        for s in unmatched[:10]:
            print("  slot", s, "values", [hex(v) for v in slot_vals[s][:6]])
        raise SystemExit(1)
    # all-zero-on-all-probes slots are suspicious (insufficient excitation)
    zero_slots = [s for s in range(nwit) if not any(slot_vals[s])]
    print(f"[{circuit}] slots identically zero over probes: {len(zero_slots)} {zero_slots[:20]}")
    os.makedirs(LAYOUT_DIR, exist_ok=True)
    path = os.path.join(LAYOUT_DIR, f"{circuit}.layout")
    with open(path, "w") as f:
        f.write(f"# witness slot layout of {cfg['wasm']} recovered by tools/recover_layout.py\n")
        f.write(f"# nwit {nwit}\n# prime {hex(prime)}\n# natoms {natoms}\n")
        s = 0
        while s < nwit:
            a, b = layout[s]
            e = s + 1
            if b < 0:
                while e < nwit and layout[e] == (a + (e - s), -1):
                    e += 1
                f.write(f"W {s} {a} {e - s}\n")
            else:
                while e < nwit and layout[e] == (a, b + (e - s)):
                    e += 1
                f.write(f"B {s} {a} {b} {e - s}\n")
            s = e
    print(f"[{circuit}] wrote {path}")
    validate(circuit, holdout, seed + 1)


def validate(circuit, n, seed):
    cfg = CIRCUITS[circuit]
    kind, prime = cfg["kind"], cfg["prime"]
    layout = parse_layout(os.path.join(LAYOUT_DIR, f"{circuit}.layout"))
    rng = random.Random(seed)
    if kind == "compression":
        probes = [compression_probe(rng, wild=(j % 3 == 2)) for j in range(n)]
    else:
        probes = [nova_probe(rng, wild=(j % 3 == 2)) for j in range(n)]
        for i in range(64):
            for bit in (0, 1):
                probes.append(nova_probe(rng, directed=(i, bit), wild=(bit == 1)))
    print(f"[{circuit}] validating on {len(probes)} held-out inputs", flush=True)
    bodies, errors = run_oracle(cfg["wasm"], probes)
    bad = 0
    for q, pr in enumerate(probes):
        if q in errors:
            try:
                eval_atoms(kind, prime, pr)
                print("  model accepted an input the WASM rejected:", json.dumps(pr, default=str), errors[q][:80])
                bad += 1
            except M.AssertFailed:
                pass
            continue
        pred = predict_body(layout, eval_atoms(kind, prime, pr))
        if pred != bodies[q].tobytes():
            bad += 1
            if bad < 4:
                got = np.frombuffer(pred, dtype=np.uint8).reshape(-1, 32)
                diff = np.nonzero((got != bodies[q].reshape(-1, 32)).any(axis=1))[0]
                print("  mismatch probe", q, "slots", diff[:10], json.dumps(pr, default=str))
    print(f"[{circuit}] held-out mismatches: {bad} / {len(probes)}")
    if bad:
        raise SystemExit(2)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("circuit", choices=list(CIRCUITS))
    ap.add_argument("--probes", type=int, default=64)
    ap.add_argument("--holdout", type=int, default=256)
    ap.add_argument("--validate-only", action="store_true")
    a = ap.parse_args()
    if a.validate_only:
        validate(a.circuit, a.holdout, 999)
    else:
        recover(a.circuit, a.probes, a.holdout)
