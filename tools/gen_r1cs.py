#!/usr/bin/env python3
"""tools/gen_r1cs.py — derive the rank-1 constraint system of `blake3_compression` (BN254, the circomkit / O1 build)
from the circuit text and write it as a standard iden3 `.r1cs` file over the witness slots of the committed WASM.

The reference checkout has no `.r1cs` (`.MISSING_LARGE_BLOBS`), but every constraint of the circuit is written out in
  /root/reference/circuits/blake3_common.circom:15-26,42-80,142-203     (Blake3Permute, XOR2, XorWord2, ToBits, Bits33, Bits34)
  /root/reference/circuits/blake3_compression.circom:17-228             (IV, RotXorBits, RotXorWordBits, HalfFunG, MixFunG,
                                                                         SingleRound, Blake3Compression)
and the build's symbol table `/root/reference/build/blake3_compression/blake3_compression.sym` names every signal and
the witness slot it kept.  This script ELABORATES the templates the way circom does — one Python function per template,
statement by statement: `signal`, `<==` (alias, constant or constraint), `===`, `<--` (no constraint) — then applies the
two simplifications of that build (signal = signal and signal = constant substitutions; linear constraints are kept),
and checks the result against the reference before writing anything:

  1. every signal the elaboration creates exists in the .sym, and vice versa (69 380 names);
  2. every alias class has exactly the slot the .sym gives its lowest-numbered member, constants have none —
     i.e. the elaboration's aliasing IS the compiler's;
  3. the reference's committed witness (build/blake3_compression/testInp/witness.wtns) satisfies every constraint.

Nothing here reads this repository's slot tables, kernels or oracle: the constraint system is an INDEPENDENT statement of
what a valid witness is (the on-device check `b3w_r1cs_check_device` evaluates Az * Bz - Cz with it).

Build container only (it reads /root/reference).  Output (committed): hot-proofs-blake3-circom_amd/constraints/
blake3_compression.r1cs.gz — iden3 r1cs v1: header, constraints (A, B, C as (wire, coefficient) lists; A*B - C = 0),
wire-to-label map (labels = .sym signal ids).
"""
import gzip
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("B3W_REFERENCE", "/root/reference")
SYM = os.path.join(REF, "build/blake3_compression/blake3_compression.sym")
WTNS = os.path.join(REF, "build/blake3_compression/testInp/witness.wtns")
OUT = os.path.join(ROOT, "hot-proofs-blake3-circom_amd", "constraints", "blake3_compression.r1cs.gz")
P = 21888242871839275222246405745257275088548364400416034343698204186575808495617      # BN254 scalar field


# ------------------------------------------------------------------ a very small circom: signals, linear combinations
class Circuit:
    def __init__(self):
        self.names = []            # signal id -> hierarchical name (circom numbers them differently; names are the key)
        self.parent = []           # union-find over signals related by `a <== b`
        self.const = {}            # class root -> constant value (signal <== constant)
        self.cons = []             # (A, B, C) with A*B - C = 0; each an LC = {signal or None (the constant 1): coefficient}

    def signal(self, name):
        self.names.append(name)
        self.parent.append(len(self.parent))
        return len(self.names) - 1

    def find(self, s):
        while self.parent[s] != s:
            self.parent[s] = self.parent[self.parent[s]]
            s = self.parent[s]
        return s

    def alias(self, a, b):
        """a <== b with b a single signal: no constraint survives, the two are one wire"""
        ra, rb = self.find(a), self.find(b)
        if ra == rb:
            return
        assert not (ra in self.const and rb in self.const)
        if ra in self.const:
            ra, rb = rb, ra
        self.parent[ra] = rb       # constants stay roots

    def assign(self, sig, lc):
        """sig <== lc for a linear right-hand side"""
        lc = {k: v % P for k, v in lc.items() if v % P}
        keys = list(lc)
        if keys == [None] or not keys:
            self.const[self.find(sig)] = lc.get(None, 0)
        elif len(keys) == 1 and lc[keys[0]] == 1:
            self.alias(sig, keys[0])
        else:
            self.linear(sub(lc, {sig: 1}))

    def linear(self, lc):
        """lc === 0"""
        self.cons.append(({}, {}, neg(lc)))       # 0 * 0 - C = 0 with C = -lc (sign is immaterial)

    def quadratic(self, a, b, c):
        """a * b === c"""
        self.cons.append((dict(a), dict(b), dict(c)))


def add(*lcs):
    out = {}
    for lc in lcs:
        for k, v in lc.items():
            out[k] = (out.get(k, 0) + v) % P
    return out


def scale(lc, f):
    return {k: v * f % P for k, v in lc.items()}


def neg(lc):
    return scale(lc, P - 1)


def sub(a, b):
    return add(a, neg(b))


def S(sig):
    return {sig: 1}


def K(v):
    return {None: v % P}


# ------------------------------------------------------------------ templates (blake3_common.circom)
def ToBits(c, pfx, n=32):
    """blake3_common.circom:142-154"""
    inp = c.signal(f"{pfx}.inp")
    out = [c.signal(f"{pfx}.out[{i}]") for i in range(n)]
    total = {}
    for i in range(n):
        c.quadratic(S(out[i]), sub(K(1), S(out[i])), {})          # out[i] * (1 - out[i]) === 0      (:150)
        total = add(total, scale(S(out[i]), 1 << i))
    c.linear(sub(S(inp), total))                                  # inp === sum                       (:153)
    return dict(inp=inp, out=out)


def BitsN(c, pfx, carries):
    """Bits33 (carries = 1, blake3_common.circom:160-178) and Bits34 (carries = 2, :183-203)"""
    inp = c.signal(f"{pfx}.inp")
    out_bits = [c.signal(f"{pfx}.out_bits[{i}]") for i in range(32)]
    out_word = c.signal(f"{pfx}.out_word")
    cs = [c.signal(f"{pfx}.{nm}") for nm in ("u", "v")[:carries]]
    total = {}
    for i in range(32):
        c.quadratic(S(out_bits[i]), sub(K(1), S(out_bits[i])), {})
        total = add(total, scale(S(out_bits[i]), 1 << i))
    for s in cs:
        c.quadratic(S(s), sub(K(1), S(s)), {})                    # u*(1-u) === 0 ; v*(1-v) === 0
    full = total
    for j, s in enumerate(cs):
        full = add(full, scale(S(s), 1 << (32 + j)))
    c.linear(sub(S(inp), full))                                   # inp === sum + 2^32 u (+ 2^33 v)
    c.assign(out_word, total)                                     # out_word <== sum
    return dict(inp=inp, out_bits=out_bits, out_word=out_word)


def XOR2(c, pfx):
    """blake3_common.circom:42-50"""
    x, y, out = c.signal(f"{pfx}.x"), c.signal(f"{pfx}.y"), c.signal(f"{pfx}.out")
    # out <== x + y - 2*x*y   ->   (2x) * y = x + y - out
    c.quadratic(scale(S(x), 2), S(y), sub(add(S(x), S(y)), S(out)))
    return dict(x=x, y=y, out=out)


def XorWord2(c, pfx, n=32):
    """blake3_common.circom:55-80"""
    x, y = c.signal(f"{pfx}.x"), c.signal(f"{pfx}.y")
    out_bits = [c.signal(f"{pfx}.out_bits[{i}]") for i in range(n)]
    out_word = c.signal(f"{pfx}.out_word")
    tb_x, tb_y = ToBits(c, f"{pfx}.tb_x", n), ToBits(c, f"{pfx}.tb_y", n)
    c.alias(tb_x["inp"], x)
    c.alias(tb_y["inp"], y)
    acc = {}
    for i in range(n):
        g = XOR2(c, f"{pfx}.xor[{i}]")
        c.alias(g["x"], tb_x["out"][i])
        c.alias(g["y"], tb_y["out"][i])
        c.alias(out_bits[i], g["out"])
        acc = add(acc, scale(S(out_bits[i]), 1 << i))
    c.assign(out_word, acc)
    return dict(x=x, y=y, out_word=out_word)


def Blake3Permute(c, pfx):
    """blake3_common.circom:15-26"""
    sigma = [2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8]
    inp = [c.signal(f"{pfx}.inp[{j}]") for j in range(16)]
    out = [c.signal(f"{pfx}.out[{j}]") for j in range(16)]
    for j in range(16):
        c.alias(out[j], inp[sigma[j]])
    return dict(inp=inp, out=out)


# ------------------------------------------------------------------ templates (blake3_compression.circom)
def IV(c, pfx):
    """:17-24"""
    iv = [0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A, 0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19]
    out = [c.signal(f"{pfx}.out[{j}]") for j in range(8)]
    for j in range(8):
        c.assign(out[j], K(iv[j]))
    return dict(out=out)


def RotXorBits(c, pfx, R):
    """:29-47"""
    inp1 = [c.signal(f"{pfx}.inp1_bits[{i}]") for i in range(32)]
    inp2 = [c.signal(f"{pfx}.inp2_bits[{i}]") for i in range(32)]
    out_bits = [c.signal(f"{pfx}.out_bits[{i}]") for i in range(32)]
    out_word = c.signal(f"{pfx}.out_word")
    aux = [c.signal(f"{pfx}.aux[{i}]") for i in range(32)]
    for i in range(32):
        # aux[i] <== inp1[i] + inp2[i] - 2 * inp1[i] * inp2[i]
        c.quadratic(scale(S(inp1[i]), 2), S(inp2[i]), sub(add(S(inp1[i]), S(inp2[i])), S(aux[i])))
    acc = {}
    for i in range(32):
        c.alias(out_bits[i], aux[(i + R) % 32])
        acc = add(acc, scale(S(out_bits[i]), 1 << i))
    c.assign(out_word, acc)
    return dict(inp1_bits=inp1, inp2_bits=inp2, out_bits=out_bits, out_word=out_word)


def RotXorWordBits(c, pfx, R):
    """:53-67"""
    inp1_word = c.signal(f"{pfx}.inp1_word")
    inp2_bits = [c.signal(f"{pfx}.inp2_bits[{i}]") for i in range(32)]
    out_bits = [c.signal(f"{pfx}.out_bits[{i}]") for i in range(32)]
    out_word = c.signal(f"{pfx}.out_word")
    tb = ToBits(c, f"{pfx}.tb", 32)
    rx = RotXorBits(c, f"{pfx}.rx", R)
    c.alias(tb["inp"], inp1_word)
    for i in range(32):
        c.alias(rx["inp1_bits"][i], tb["out"][i])
        c.alias(rx["inp2_bits"][i], inp2_bits[i])
        c.alias(out_bits[i], rx["out_bits"][i])
    c.alias(out_word, rx["out_word"])
    return dict(inp1_word=inp1_word, inp2_bits=inp2_bits, out_bits=out_bits, out_word=out_word)


def HalfFunG(c, pfx, a, b, cc, d, R1, R2):
    """:72-100"""
    v = [c.signal(f"{pfx}.v[{i}]") for i in range(16)]
    xy = c.signal(f"{pfx}.xy")
    out = [c.signal(f"{pfx}.out[{i}]") for i in range(16)]
    for i in range(16):
        if i not in (a, b, cc, d):
            c.alias(out[i], v[i])
    add1 = BitsN(c, f"{pfx}.add1", 2)
    add3 = BitsN(c, f"{pfx}.add3", 1)
    rxor2 = RotXorWordBits(c, f"{pfx}.rxor2", R1)
    rxor4 = RotXorWordBits(c, f"{pfx}.rxor4", R2)
    c.assign(add1["inp"], add(S(v[a]), S(v[b]), S(xy)))             # add1.inp <== v[a] + v[b] + xy
    c.alias(rxor2["inp1_word"], v[d])
    for i in range(32):
        c.alias(rxor2["inp2_bits"][i], add1["out_bits"][i])
    c.assign(add3["inp"], add(S(v[cc]), S(rxor2["out_word"])))      # add3.inp <== v[c] + rxor2.out_word
    c.alias(rxor4["inp1_word"], v[b])
    for i in range(32):
        c.alias(rxor4["inp2_bits"][i], add3["out_bits"][i])
    c.alias(out[a], add1["out_word"])
    c.alias(out[d], rxor2["out_word"])
    c.alias(out[cc], add3["out_word"])
    c.alias(out[b], rxor4["out_word"])
    return dict(v=v, xy=xy, out=out)


def MixFunG(c, pfx, a, b, cc, d):
    """:106-123"""
    inp = [c.signal(f"{pfx}.inp[{i}]") for i in range(16)]
    out = [c.signal(f"{pfx}.out[{i}]") for i in range(16)]
    x, y = c.signal(f"{pfx}.x"), c.signal(f"{pfx}.y")
    h1 = HalfFunG(c, f"{pfx}.half1", a, b, cc, d, 16, 12)
    h2 = HalfFunG(c, f"{pfx}.half2", a, b, cc, d, 8, 7)
    for i in range(16):
        c.alias(h1["v"][i], inp[i])
    c.alias(h1["xy"], x)
    for i in range(16):
        c.alias(h2["v"][i], h1["out"][i])
    c.alias(h2["xy"], y)
    for i in range(16):
        c.alias(out[i], h2["out"][i])
    return dict(inp=inp, out=out, x=x, y=y)


def SingleRound(c, pfx):
    """:128-161"""
    inp = [c.signal(f"{pfx}.inp[{i}]") for i in range(16)]
    msg = [c.signal(f"{pfx}.msg[{i}]") for i in range(16)]
    out = [c.signal(f"{pfx}.out[{i}]") for i in range(16)]
    vs = [[c.signal(f"{pfx}.vs[{k}][{i}]") for i in range(16)] for k in range(9)]
    for i in range(16):
        c.alias(vs[0][i], inp[i])
    idx = [(0, 4, 8, 12), (1, 5, 9, 13), (2, 6, 10, 14), (3, 7, 11, 15), (0, 5, 10, 15), (1, 6, 11, 12), (2, 7, 8, 13), (3, 4, 9, 14)]
    GS = []
    for g in range(8):
        G = MixFunG(c, f"{pfx}.GS[{g}]", *idx[g])
        c.alias(G["x"], msg[2 * g])
        c.alias(G["y"], msg[2 * g + 1])
        GS.append(G)
    for g in range(8):
        for i in range(16):
            c.alias(GS[g]["inp"][i], vs[g][i])
            c.alias(vs[g + 1][i], GS[g]["out"][i])
    for i in range(16):
        c.alias(out[i], vs[8][i])
    return dict(inp=inp, msg=msg, out=out)


def Blake3Compression(c, pfx="main"):
    """:171-228"""
    h = [c.signal(f"{pfx}.h[{i}]") for i in range(8)]
    m = [c.signal(f"{pfx}.m[{i}]") for i in range(16)]
    t = [c.signal(f"{pfx}.t[{i}]") for i in range(2)]
    b, d = c.signal(f"{pfx}.b"), c.signal(f"{pfx}.d")
    out = [c.signal(f"{pfx}.out[{i}]") for i in range(16)]
    init = [c.signal(f"{pfx}.init[{i}]") for i in range(16)]
    iv = IV(c, f"{pfx}.iv")
    for i in range(8):
        c.alias(init[i], h[i])
    for i in range(4):
        c.alias(init[i + 8], iv["out"][i])
    for i in range(2):
        c.alias(init[i + 12], t[i])
    c.alias(init[14], b)
    c.alias(init[15], d)
    rounds = [SingleRound(c, f"{pfx}.rounds[0]")]
    for i in range(16):
        c.alias(rounds[0]["msg"][i], m[i])
        c.alias(rounds[0]["inp"][i], init[i])
    perms = []
    for i in range(6):
        rounds.append(SingleRound(c, f"{pfx}.rounds[{i + 1}]"))
        perms.append(Blake3Permute(c, f"{pfx}.permuters[{i}]"))
        src = m if i == 0 else perms[i - 1]["out"]
        for j in range(16):
            c.alias(perms[i]["inp"][j], src[j])
            c.alias(rounds[i + 1]["msg"][j], perms[i]["out"][j])
            c.alias(rounds[i + 1]["inp"][j], rounds[i]["out"][j])
    for i in range(16):
        X = XorWord2(c, f"{pfx}.outXor[{i}]", 32)
        c.alias(X["x"], rounds[6]["out"][i])
        c.alias(X["y"], rounds[6]["out"][i + 8] if i < 8 else h[i - 8])
        c.alias(out[i], X["out_word"])
    return dict(h=h, m=m, t=t, b=b, d=d, out=out)


# ------------------------------------------------------------------ against the reference's symbol table and witness
def read_sym(path):
    """id,witnessIdx,componentId,name  ->  {name: (id, witnessIdx)}"""
    tab = {}
    for line in open(path):
        i, w, _, name = line.rstrip("\n").split(",", 3)
        tab[name] = (int(i), int(w))
    return tab


def read_wtns(path):
    raw = open(path, "rb").read()
    assert raw[:4] == b"wtns"
    n = struct.unpack_from("<I", raw, 60)[0]
    return [int.from_bytes(raw[76 + 32 * i: 108 + 32 * i], "little") for i in range(n)]


def lower(c, sym):
    """Substitute every signal by its wire (the slot of its alias class) or its constant; check the classes against
    the .sym.  Returns (constraints over wires, nWires, wire2label)."""
    assert set(c.names) == set(sym), (len(c.names), len(sym), sorted(set(sym) - set(c.names))[:5], sorted(set(c.names) - set(sym))[:5])
    assert len(set(c.names)) == len(c.names)
    classes = {}
    for s in range(len(c.names)):
        classes.setdefault(c.find(s), []).append(s)
    wire_of, label_of = {}, {0: 0}
    for root, members in classes.items():
        kept = [(sym[c.names[s]][0], sym[c.names[s]][1]) for s in members]
        slots = [w for _, w in kept if w >= 0]
        if root in c.const:
            assert not slots, ("a constant signal kept a witness slot", c.names[root])
            continue
        lowest = min(kept)
        assert len(slots) == 1 and lowest[1] == slots[0], ("alias class disagrees with the .sym", [c.names[s] for s in members][:4], kept[:4])
        wire_of[root] = slots[0]
        assert slots[0] not in label_of
        label_of[slots[0]] = lowest[0]
    nwires = 1 + len(wire_of)
    assert sorted(label_of) == list(range(nwires)), "witness slots are not 0 .. nWires-1"

    def low(lc):
        out = {}
        for k, v in lc.items():
            if k is None:
                w, f = 0, v
            else:
                r = c.find(k)
                w, f = (0, v * c.const[r]) if r in c.const else (wire_of[r], v)
            out[w] = (out.get(w, 0) + f) % P
        return {w: f for w, f in sorted(out.items()) if f}
    cons = [(low(a), low(b), low(cc)) for a, b, cc in c.cons]
    return cons, nwires, [label_of[w] for w in range(nwires)]


def violated(cons, z):
    dot = lambda lc: sum(f * z[w] for w, f in lc.items()) % P
    return [i for i, (a, b, cc) in enumerate(cons) if (dot(a) * dot(b) - dot(cc)) % P]


def write_r1cs(path, cons, nwires, wire2label, n_pub_out, n_pub_in, n_prv_in, n_labels):
    """iden3 r1cs binary format, version 1 (https://github.com/iden3/r1csfile/blob/master/doc/r1cs_bin_format.md):
    "r1cs" | u32 version | u32 nSections | { u32 type | u64 size | body }*
    type 1 header: u32 fieldSize | prime | u32 nWires nPubOut nPubIn nPrvIn | u64 nLabels | u32 mConstraints
    type 2 constraints: per constraint A, B, C, each u32 n | n * (u32 wire | fieldSize-byte LE coefficient)
    type 3 wire2label: nWires * u64"""
    def lc_bytes(lc):
        return struct.pack("<I", len(lc)) + b"".join(struct.pack("<I", w) + f.to_bytes(32, "little") for w, f in lc.items())
    header = struct.pack("<I", 32) + P.to_bytes(32, "little") + struct.pack("<IIIIQI", nwires, n_pub_out, n_pub_in, n_prv_in, n_labels, len(cons))
    body = b"".join(lc_bytes(a) + lc_bytes(b) + lc_bytes(cc) for a, b, cc in cons)
    w2l = b"".join(struct.pack("<Q", x) for x in wire2label)
    blob = b"r1cs" + struct.pack("<II", 1, 3)
    for typ, sec in ((1, header), (2, body), (3, w2l)):
        blob += struct.pack("<IQ", typ, len(sec)) + sec
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "wb") as raw, gzip.GzipFile(filename="", fileobj=raw, mode="wb", mtime=0) as f:
        f.write(blob)                                    # no name, no time stamp: the file is reproducible byte for byte
    return len(blob)


def main():
    out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else OUT
    sym = read_sym(SYM)
    c = Circuit()
    Blake3Compression(c)
    cons, nwires, w2l = lower(c, sym)
    z = read_wtns(WTNS)
    assert len(z) == nwires == 24093, (len(z), nwires)
    bad = violated(cons, z)
    assert not bad, f"the reference's own witness violates {len(bad)} derived constraints, first {bad[:5]}"
    kinds = {"bool": 0, "quadratic": 0, "linear": 0}
    for a, b, cc in cons:
        kinds["linear" if not a else "bool" if not cc else "quadratic"] += 1
    nnz = sum(len(a) + len(b) + len(cc) for a, b, cc in cons)
    size = write_r1cs(out, cons, nwires, w2l, 16, 0, 28, max(i for i, _ in sym.values()) + 1)
    print(f"{len(c.names)} signals = the .sym's; {nwires} wires; {len(cons)} constraints {kinds}; {nnz} non-zeros; "
          f"reference witness satisfies all; wrote {out} ({size} bytes before gzip)")


if __name__ == "__main__":
    sys.exit(main())
