#!/usr/bin/env python3
"""tools/gen_r1cs.py [--circuit compression | nova_o1] — derive the rank-1 constraint system of `blake3_compression`
(BN254, the circomkit / O1 build) or of the circomkit build of `blake3_nova` (BN254, 24 614 wires) from the circuit text and
write it as a standard iden3 `.r1cs` file over the witness slots of the committed WASM.

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

`blake3_nova` has no .sym in the checkout (`.MISSING_LARGE_BLOBS`), so its wires come from circom's numbering RULE instead,
which this script implements and first proves on blake3_compression, where the .sym exists: inside a template instance the
outputs, then the inputs, then the intermediate signals get consecutive ids in declaration order, then its sub-components
follow in alphabetical order (arrays by index), recursively; an alias class keeps its lowest id; witness slots are handed
out in increasing id order over the classes that are not constants (`rule ids == .sym ids` and `rule slots == .sym slots`
for all 69 380 signals).  The nova templates are elaborated as the committed WASMs were compiled — WITHOUT the two
Num2Bits(8) range checks of circuits/blake3_nova.circom:25-29 (they are in no committed WASM) — with
circomlib 2.0.5's IsZero / IsEqual / Num2Bits / LessThan / GreaterEqThan / NOT / AND / OR (yarn.lock:1243; not vendored:
restated from the published library).  Checks for nova: exactly 24 614 wires (the WASM's getWitnessSize), and every witness
the reference WASM produces for the accepted golden inputs (tests/golden/nova_bn254_o1.json, non-canonical ones included;
run here through tools/wasm_oracle.js) satisfies every constraint.

Build container only (it reads /root/reference).  Output (committed): hot-proofs-blake3-circom_amd/constraints/
blake3_compression.r1cs.gz, blake3_nova_bn254_o1.r1cs.gz — iden3 r1cs v1: header, constraints (A, B, C as (wire,
coefficient) lists; A*B - C = 0), wire-to-label map (labels = circom signal ids).
"""
import gzip
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("B3W_REFERENCE", "/root/reference")
SYM = os.path.join(REF, "build/blake3_compression/blake3_compression.sym")
WTNS = os.path.join(REF, "build/blake3_compression/testInp/witness.wtns")
OUT = os.path.join(ROOT, "hot-proofs-blake3-circom_amd", "constraints", "blake3_compression.r1cs.gz")
P = 21888242871839275222246405745257275088548364400416034343698204186575808495617      # BN254 scalar field


# ------------------------------------------------------------------ a very small circom: signals, linear combinations
class Circuit:
    def __init__(self):
        self.names = []            # signal -> hierarchical name
        self.kinds = []            # 0 output, 1 input, 2 intermediate (of its template instance)
        self.parent = []           # union-find over signals the build's simplification merges
        self.const = {}            # class root -> constant value
        self.cons = []             # (A, B, C) with A*B - C = 0; each an LC = {signal or None (the constant 1): coefficient}

    def signal(self, name, kind):
        self.names.append(name)
        self.kinds.append(kind)
        self.parent.append(len(self.parent))
        return len(self.names) - 1

    def out(self, name):
        return self.signal(name, 0)

    def inp(self, name):
        return self.signal(name, 1)

    def mid(self, name):
        return self.signal(name, 2)

    def find(self, s):
        while self.parent[s] != s:
            self.parent[s] = self.parent[self.parent[s]]
            s = self.parent[s]
        return s

    def alias(self, a, b):
        """a <== b with b a single signal: the constraint a - b = 0, which the build's simplification turns into "one wire"
        (see simplify) — recorded as a constraint like any other so that the classification sees it"""
        self.linear(sub(S(a), S(b)))

    def union(self, a, b):
        ra, rb = self.find(a), self.find(b)
        if ra != rb:
            self.parent[ra] = rb

    def assign(self, sig, lc):
        """sig <== lc for a linear right-hand side"""
        self.linear(sub(lc, {sig: 1}))

    def linear(self, lc):
        """lc === 0"""
        self.cons.append(({}, {}, neg(lc)))       # 0 * 0 - C = 0 with C = -lc (sign is immaterial)

    def quadratic(self, a, b, c):
        """a * b === c"""
        self.cons.append((dict(a), dict(b), dict(c)))

    def resolve(self, lc):
        """lc over class representatives, constants folded into the None term"""
        out = {}
        for k, v in lc.items():
            if k is not None:
                r = self.find(k)
                k, v = (None, v * self.const[r]) if r in self.const else (r, v)
            out[k] = (out.get(k, 0) + v) % P
        return {k: v for k, v in out.items() if v}

    def simplify(self, public=()):
        """The two simplifications of the circomkit (O1) build.  The constraints AS WRITTEN are classified once: `signal =
        signal` (two signals, coefficients k and -k, no constant) merges the two into one wire; `signal = constant` (one
        signal) makes it a constant; both constraints disappear.  Everything else stays — linear constraints included,
        also those that shrink to one or two signals only after the substitutions (exceed_depth.lt.n2b.out[8] = 1 stays a
        wire with its constraint).  `public` = the main component's outputs and public inputs: they always keep their own
        wire, so an equality between two of them stays a constraint (n_blocks_out = n_blocks)."""
        public = set(public)
        eqs, consts, keep = [], [], []
        for a, b, cc in self.cons:
            if a or b:
                keep.append((a, b, cc))
                continue
            lc = {k: v % P for k, v in cc.items() if v % P}
            sigs = [k for k in lc if k is not None]
            if len(sigs) == 2 and None not in lc and (lc[sigs[0]] + lc[sigs[1]]) % P == 0:
                eqs.append((sigs[0], sigs[1], (a, b, cc)))
            elif len(sigs) == 1:
                consts.append((sigs[0], (-lc.get(None, 0)) * pow(lc[sigs[0]], -1, P) % P))
            else:
                keep.append((a, b, cc))
        has_public = {}
        for s in public:
            has_public[s] = True
        for x, y, con in eqs:
            rx, ry = self.find(x), self.find(y)
            if rx == ry:
                continue
            if has_public.get(rx) and has_public.get(ry):
                keep.append(con)                              # two public wires: the equality stays a constraint
                continue
            self.union(x, y)
            if has_public.get(rx) or has_public.get(ry):
                has_public[self.find(x)] = True
        for s0, val in consts:
            r = self.find(s0)
            assert not has_public.get(r), "a public signal assigned a constant: not handled"
            assert self.const.get(r, val) == val
            self.const[r] = val
        self.cons = keep

    # ---- circom's signal numbering
    def number(self):
        """{signal: id}: per template instance outputs, inputs, intermediates in declaration order, then the
        sub-components in alphabetical order (component arrays by index), recursively; id 0 is the constant one."""
        tree = {}
        for s, name in enumerate(self.names):
            comp, _ = name.rsplit(".", 1)
            node = tree
            for part in comp.split("."):
                node = node.setdefault(("c", part), {})
            node.setdefault(("s",), []).append(s)
        ids, nxt = {}, [1]

        def order_key(part):
            mt = re.match(r"^(\w+?)((?:\[\d+\])*)$", part)
            return (mt.group(1), [int(x) for x in re.findall(r"\[(\d+)\]", mt.group(2))])

        def walk(node):
            own = node.get(("s",), [])
            for kind in (0, 1, 2):
                for s in own:                                 # creation order = declaration order within a kind
                    if self.kinds[s] == kind:
                        ids[s] = nxt[0]
                        nxt[0] += 1
            for key in sorted((k for k in node if k[0] == "c"), key=lambda k: order_key(k[1])):
                walk(node[key])
        walk(tree)
        return ids


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
# Signals are created in the source's declaration order: the numbering rule depends on it.
def ToBits(c, pfx, n=32):
    """blake3_common.circom:142-154"""
    inp = c.inp(f"{pfx}.inp")
    out = [c.out(f"{pfx}.out[{i}]") for i in range(n)]
    total = {}
    for i in range(n):
        c.quadratic(S(out[i]), sub(K(1), S(out[i])), {})          # out[i] * (1 - out[i]) === 0      (:150)
        total = add(total, scale(S(out[i]), 1 << i))
    c.linear(sub(S(inp), total))                                  # inp === sum                       (:153)
    return dict(inp=inp, out=out)


def BitsN(c, pfx, carries):
    """Bits33 (carries = 1, blake3_common.circom:160-178) and Bits34 (carries = 2, :183-203)"""
    inp = c.inp(f"{pfx}.inp")
    out_bits = [c.out(f"{pfx}.out_bits[{i}]") for i in range(32)]
    out_word = c.out(f"{pfx}.out_word")
    cs = [c.mid(f"{pfx}.{nm}") for nm in ("u", "v")[:carries]]
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
    x, y, out = c.inp(f"{pfx}.x"), c.inp(f"{pfx}.y"), c.out(f"{pfx}.out")
    # out <== x + y - 2*x*y   ->   (2x) * y = x + y - out
    c.quadratic(scale(S(x), 2), S(y), sub(add(S(x), S(y)), S(out)))
    return dict(x=x, y=y, out=out)


def XorWord2(c, pfx, n=32):
    """blake3_common.circom:55-80"""
    x, y = c.inp(f"{pfx}.x"), c.inp(f"{pfx}.y")
    out_bits = [c.mid(f"{pfx}.out_bits[{i}]") for i in range(n)]
    out_word = c.out(f"{pfx}.out_word")
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
    inp = [c.inp(f"{pfx}.inp[{j}]") for j in range(16)]
    out = [c.out(f"{pfx}.out[{j}]") for j in range(16)]
    for j in range(16):
        c.alias(out[j], inp[sigma[j]])
    return dict(inp=inp, out=out)


# ------------------------------------------------------------------ templates (blake3_compression.circom)
IV_WORDS = [0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A, 0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19]


def IV(c, pfx):
    """:17-24"""
    out = [c.out(f"{pfx}.out[{j}]") for j in range(8)]
    for j in range(8):
        c.assign(out[j], K(IV_WORDS[j]))
    return dict(out=out)


def RotXorBits(c, pfx, R):
    """:29-47"""
    inp1 = [c.inp(f"{pfx}.inp1_bits[{i}]") for i in range(32)]
    inp2 = [c.inp(f"{pfx}.inp2_bits[{i}]") for i in range(32)]
    out_bits = [c.out(f"{pfx}.out_bits[{i}]") for i in range(32)]
    out_word = c.out(f"{pfx}.out_word")
    aux = [c.mid(f"{pfx}.aux[{i}]") for i in range(32)]
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
    inp1_word = c.inp(f"{pfx}.inp1_word")
    inp2_bits = [c.inp(f"{pfx}.inp2_bits[{i}]") for i in range(32)]
    out_bits = [c.out(f"{pfx}.out_bits[{i}]") for i in range(32)]
    out_word = c.out(f"{pfx}.out_word")
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
    v = [c.inp(f"{pfx}.v[{i}]") for i in range(16)]
    xy = c.inp(f"{pfx}.xy")
    out = [c.out(f"{pfx}.out[{i}]") for i in range(16)]
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
    inp = [c.inp(f"{pfx}.inp[{i}]") for i in range(16)]
    out = [c.out(f"{pfx}.out[{i}]") for i in range(16)]
    x, y = c.inp(f"{pfx}.x"), c.inp(f"{pfx}.y")
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
    inp = [c.inp(f"{pfx}.inp[{i}]") for i in range(16)]
    msg = [c.inp(f"{pfx}.msg[{i}]") for i in range(16)]
    out = [c.out(f"{pfx}.out[{i}]") for i in range(16)]
    vs = [[c.mid(f"{pfx}.vs[{k}][{i}]") for i in range(16)] for k in range(9)]
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
    h = [c.inp(f"{pfx}.h[{i}]") for i in range(8)]
    m = [c.inp(f"{pfx}.m[{i}]") for i in range(16)]
    t = [c.inp(f"{pfx}.t[{i}]") for i in range(2)]
    b, d = c.inp(f"{pfx}.b"), c.inp(f"{pfx}.d")
    out = [c.out(f"{pfx}.out[{i}]") for i in range(16)]
    init = [c.mid(f"{pfx}.init[{i}]") for i in range(16)]
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


# ------------------------------------------------------------------ circomlib 2.0.5 (comparators, gates, bitify)
def Num2Bits(c, pfx, n):
    """bitify.circom Num2Bits(n): out[i] <-- (in >> i) & 1; out[i] * (out[i] - 1) === 0; sum 2^i out[i] === in"""
    inp = c.inp(f"{pfx}.in")
    out = [c.out(f"{pfx}.out[{i}]") for i in range(n)]
    lc1 = {}
    for i in range(n):
        c.quadratic(S(out[i]), sub(S(out[i]), K(1)), {})
        lc1 = add(lc1, scale(S(out[i]), 1 << i))
    c.linear(sub(lc1, S(inp)))
    return {"in": inp, "out": out}


def IsZero(c, pfx):
    """comparators.circom IsZero: inv <-- in != 0 ? 1/in : 0; out <== -in*inv + 1; in*out === 0"""
    inp, out, inv = c.inp(f"{pfx}.in"), c.out(f"{pfx}.out"), c.mid(f"{pfx}.inv")
    c.quadratic(neg(S(inp)), S(inv), sub(S(out), K(1)))           # (-in) * inv = out - 1
    c.quadratic(S(inp), S(out), {})
    return {"in": inp, "out": out}


def IsEqual(c, pfx):
    """comparators.circom IsEqual: isz.in <== in[1] - in[0]; out <== isz.out"""
    inp = [c.inp(f"{pfx}.in[{i}]") for i in range(2)]
    out = c.out(f"{pfx}.out")
    isz = IsZero(c, f"{pfx}.isz")
    c.assign(isz["in"], sub(S(inp[1]), S(inp[0])))
    c.alias(out, isz["out"])
    return {"in": inp, "out": out}


def LessThan(c, pfx, n):
    """comparators.circom LessThan(n): n2b = Num2Bits(n+1); n2b.in <== in[0] + (1<<n) - in[1]; out <== 1 - n2b.out[n]"""
    inp = [c.inp(f"{pfx}.in[{i}]") for i in range(2)]
    out = c.out(f"{pfx}.out")
    n2b = Num2Bits(c, f"{pfx}.n2b", n + 1)
    c.assign(n2b["in"], sub(add(S(inp[0]), K(1 << n)), S(inp[1])))
    c.assign(out, sub(K(1), S(n2b["out"][n])))
    return {"in": inp, "out": out}


def GreaterEqThan(c, pfx, n):
    """comparators.circom GreaterEqThan(n): lt = LessThan(n); lt.in[0] <== in[1]; lt.in[1] <== in[0] + 1; out <== lt.out"""
    inp = [c.inp(f"{pfx}.in[{i}]") for i in range(2)]
    out = c.out(f"{pfx}.out")
    lt = LessThan(c, f"{pfx}.lt", n)
    c.alias(lt["in"][0], inp[1])
    c.assign(lt["in"][1], add(S(inp[0]), K(1)))
    c.alias(out, lt["out"])
    return {"in": inp, "out": out}


def NOT(c, pfx):
    """gates.circom NOT: out <== 1 + in - 2*in"""
    inp, out = c.inp(f"{pfx}.in"), c.out(f"{pfx}.out")
    c.assign(out, sub(K(1), S(inp)))
    return {"in": inp, "out": out}


def AND(c, pfx):
    """gates.circom AND: out <== a*b"""
    a, b, out = c.inp(f"{pfx}.a"), c.inp(f"{pfx}.b"), c.out(f"{pfx}.out")
    c.quadratic(S(a), S(b), S(out))
    return dict(a=a, b=b, out=out)


def OR(c, pfx):
    """gates.circom OR: out <== a + b - a*b"""
    a, b, out = c.inp(f"{pfx}.a"), c.inp(f"{pfx}.b"), c.out(f"{pfx}.out")
    c.quadratic(S(a), S(b), sub(add(S(a), S(b)), S(out)))
    return dict(a=a, b=b, out=out)


# ------------------------------------------------------------------ templates (blake3_nova.circom), as the WASMs were compiled
def CheckDepth(c, pfx):
    """Blake3NovaTreePath_CheckDepth :13-45 WITHOUT the Num2Bits(8) checks of :25-29 (absent from every committed WASM)"""
    depth, leaf_depth = c.inp(f"{pfx}.depth"), c.inp(f"{pfx}.leaf_depth")
    is_root, is_parent = c.out(f"{pfx}.is_root"), c.out(f"{pfx}.is_parent")
    check_root = IsEqual(c, f"{pfx}.check_root")
    c.alias(check_root["in"][0], depth)
    c.assign(check_root["in"][1], K(0))
    c.alias(is_root, check_root["out"])
    check_parent = LessThan(c, f"{pfx}.check_parent", 8)
    c.alias(check_parent["in"][0], depth)
    c.assign(check_parent["in"][1], sub(S(leaf_depth), K(1)))
    c.alias(is_parent, check_parent["out"])
    exceed = GreaterEqThan(c, f"{pfx}.exceed_depth", 8)
    c.alias(exceed["in"][0], depth)
    c.alias(exceed["in"][1], leaf_depth)
    c.linear(S(exceed["out"]))                                    # exceed_depth.out === 0
    return dict(depth=depth, leaf_depth=leaf_depth, is_root=is_root, is_parent=is_parent)


def GetDownLeftPath(c, pfx):
    """Blake3GetDownLeftPath :47-84"""
    depth, leaf_idx = c.inp(f"{pfx}.depth"), c.inp(f"{pfx}.leaf_idx")
    is_parent, total_depth = c.inp(f"{pfx}.is_parent"), c.inp(f"{pfx}.total_depth")
    out = c.out(f"{pfx}.out")
    n2b = Num2Bits(c, f"{pfx}.n2b", 65)
    c.alias(n2b["in"], leaf_idx)
    bit_at_depth = [c.mid(f"{pfx}.bit_at_depth[{i}]") for i in range(65)]
    for i in range(64):
        eq = IsEqual(c, f"{pfx}.eqs[{i}]")
        c.alias(eq["in"][0], depth)
        c.assign(eq["in"][1], sub(S(total_depth), K(i + 2)))
        # bit_at_depth[i] <== bit_at_depth[i-1] + (1 - n2b.out[i]) * eqs[i].out
        prev = S(bit_at_depth[i - 1]) if i else {}
        c.quadratic(sub(K(1), S(n2b["out"][i])), S(eq["out"]), sub(S(bit_at_depth[i]), prev))
    # out <== 1 * (1 - is_parent) + 1 * is_parent * bit_at_depth[63]
    c.quadratic(S(is_parent), S(bit_at_depth[63]), sub(add(S(out), S(is_parent)), K(1)))
    c.quadratic(S(out), sub(K(1), S(out)), {})
    return dict(depth=depth, leaf_idx=leaf_idx, is_parent=is_parent, total_depth=total_depth, out=out)


def GetFinal_m(c, pfx):
    """Blake3GetFinal_m :86-120"""
    h = [c.inp(f"{pfx}.h[{i}]") for i in range(8)]
    m = [c.inp(f"{pfx}.m[{i}]") for i in range(16)]
    is_parent, depth = c.inp(f"{pfx}.is_parent"), c.inp(f"{pfx}.depth")
    total_depth, chunk_idx = c.inp(f"{pfx}.total_depth"), c.inp(f"{pfx}.chunk_idx")
    out_m = [c.out(f"{pfx}.out_m[{i}]") for i in range(16)]
    dlp = GetDownLeftPath(c, f"{pfx}.down_left_path")
    c.alias(dlp["depth"], depth)
    c.alias(dlp["leaf_idx"], chunk_idx)
    c.alias(dlp["is_parent"], is_parent)
    c.alias(dlp["total_depth"], total_depth)
    m_is_parent = [c.mid(f"{pfx}.m_is_parent[{i}]") for i in range(16)]
    tmp_down = [c.mid(f"{pfx}.tmp_down[{i}]") for i in range(16)]
    tmp_is_par = [c.mid(f"{pfx}.tmp_is_par[{i}]") for i in range(16)]
    dl, ndl = S(dlp["out"]), sub(K(1), S(dlp["out"]))
    for i in range(16):
        if i < 8:
            c.quadratic(S(h[i]), dl, S(tmp_down[i]))
            c.quadratic(S(m[i]), ndl, sub(S(m_is_parent[i]), S(tmp_down[i])))
        else:
            c.quadratic(S(h[i - 8]), ndl, S(tmp_down[i]))
            c.quadratic(S(m[i - 8]), dl, sub(S(m_is_parent[i]), S(tmp_down[i])))
        c.quadratic(S(m_is_parent[i]), S(is_parent), S(tmp_is_par[i]))
        c.quadratic(S(m[i]), sub(K(1), S(is_parent)), sub(S(out_m[i]), S(tmp_is_par[i])))
    return dict(h=h, m=m, is_parent=is_parent, depth=depth, total_depth=total_depth, chunk_idx=chunk_idx, out_m=out_m)


def GetFlag(c, pfx, D_FLAGS):
    """Blake3GetFlag :122-167"""
    is_parent, is_root = c.inp(f"{pfx}.is_parent"), c.inp(f"{pfx}.is_root")
    block_count, n_blocks = c.inp(f"{pfx}.block_count"), c.inp(f"{pfx}.n_blocks")
    out, is_last_block = c.out(f"{pfx}.out"), c.out(f"{pfx}.is_last_block")
    use_root_flag = c.mid(f"{pfx}.use_root_flag")
    not_root, not_parent = NOT(c, f"{pfx}.not_root"), NOT(c, f"{pfx}.not_parent")
    c.alias(not_root["in"], is_root)
    c.alias(not_parent["in"], is_parent)
    cbc = [IsEqual(c, f"{pfx}.check_block_counts[{i}]") for i in range(2)]
    c.alias(cbc[0]["in"][0], block_count)
    c.assign(cbc[0]["in"][1], K(0))
    c.alias(cbc[1]["in"][0], block_count)
    c.assign(cbc[1]["in"][1], sub(S(n_blocks), K(1)))
    c.quadratic(S(cbc[1]["out"]), S(not_parent["out"]), S(is_last_block))
    first, last = AND(c, f"{pfx}.first_block_flag_set"), AND(c, f"{pfx}.last_block_flag_set")
    c.alias(first["a"], cbc[0]["out"]); c.alias(first["b"], not_parent["out"])
    c.alias(last["a"], cbc[1]["out"]); c.alias(last["b"], not_parent["out"])
    urt = OR(c, f"{pfx}.use_root_flag_tmp")
    c.alias(urt["a"], is_parent); c.alias(urt["b"], cbc[1]["out"])
    c.quadratic(S(urt["out"]), S(is_root), S(use_root_flag))
    c.assign(out, add(K(D_FLAGS), S(first["out"]), scale(S(last["out"]), 2), scale(S(use_root_flag), 8), scale(S(is_parent), 4)))
    return dict(is_parent=is_parent, is_root=is_root, block_count=block_count, n_blocks=n_blocks, out=out, is_last_block=is_last_block)


def Blake3Nova(c, pfx="main", D_FLAGS=0):
    """Blake3Nova(0) :169-267"""
    n_blocks, block_count = c.inp(f"{pfx}.n_blocks"), c.inp(f"{pfx}.block_count")
    h = [c.inp(f"{pfx}.h[{i}]") for i in range(8)]
    cil, cih = c.inp(f"{pfx}.chunk_idx_low"), c.inp(f"{pfx}.chunk_idx_high")
    leaf_depth, total_depth, depth = c.inp(f"{pfx}.leaf_depth"), c.inp(f"{pfx}.total_depth"), c.inp(f"{pfx}.depth")
    m = [c.inp(f"{pfx}.m[{i}]") for i in range(16)]
    b = c.inp(f"{pfx}.b")
    n_blocks_out, block_count_out = c.out(f"{pfx}.n_blocks_out"), c.out(f"{pfx}.block_count_out")
    h_out = [c.out(f"{pfx}.h_out[{i}]") for i in range(8)]
    total_depth_out, depth_out = c.out(f"{pfx}.total_depth_out"), c.out(f"{pfx}.depth_out")
    cil_out, cih_out = c.out(f"{pfx}.chunk_idx_low_out"), c.out(f"{pfx}.chunk_idx_high_out")
    leaf_depth_out = c.out(f"{pfx}.leaf_depth_out")
    tmpIV = [c.mid(f"{pfx}.tmpIV[{i}]") for i in range(8)]
    h_compression = [c.mid(f"{pfx}.h_compression[{i}]") for i in range(8)]
    decr_depth = c.mid(f"{pfx}.decr_depth")
    cd = CheckDepth(c, f"{pfx}.check_depth")
    c.alias(cd["depth"], depth)
    c.alias(cd["leaf_depth"], leaf_depth)
    fl = GetFlag(c, f"{pfx}.comp_d", D_FLAGS)
    c.alias(fl["is_parent"], cd["is_parent"])
    c.alias(fl["is_root"], cd["is_root"])
    c.alias(fl["block_count"], block_count)
    c.alias(fl["n_blocks"], n_blocks)
    iv = IV(c, f"{pfx}.iv")
    fm = GetFinal_m(c, f"{pfx}.final_m")
    for i in range(8):
        c.alias(fm["h"][i], h[i])
    for i in range(16):
        c.alias(fm["m"][i], m[i])
    c.alias(fm["is_parent"], cd["is_parent"])
    c.alias(fm["depth"], depth)
    c.alias(fm["total_depth"], total_depth)
    c.assign(fm["chunk_idx"], add(S(cil), scale(S(cih), 1 << 32)))
    npar = sub(K(1), S(cd["is_parent"]))
    for i in range(8):
        c.quadratic(S(iv["out"][i]), S(cd["is_parent"]), S(tmpIV[i]))
        c.quadratic(S(h[i]), npar, sub(S(h_compression[i]), S(tmpIV[i])))
    comp = Blake3Compression(c, f"{pfx}.blake3Compression")
    for i in range(16):
        c.alias(comp["m"][i], fm["out_m"][i])
    for i in range(8):
        c.alias(comp["h"][i], h_compression[i])
    c.alias(comp["d"], fl["out"])
    c.alias(comp["b"], b)
    c.quadratic(S(cih), npar, S(comp["t"][1]))
    c.quadratic(S(cil), npar, S(comp["t"][0]))
    for i in range(8):
        c.alias(h_out[i], comp["out"][i])
    c.assign(block_count_out, add(S(block_count), npar))
    c.alias(n_blocks_out, n_blocks)
    cdd = OR(c, f"{pfx}.check_decr_depth")
    c.alias(cdd["a"], fl["is_last_block"])
    c.alias(cdd["b"], cd["is_parent"])
    c.quadratic(S(cdd["out"]), sub(K(1), S(cd["is_root"])), S(decr_depth))
    c.quadratic(S(decr_depth), sub(K(1), S(decr_depth)), {})
    c.assign(depth_out, sub(S(depth), S(decr_depth)))
    c.alias(total_depth_out, total_depth)
    c.alias(cil_out, cil)
    c.alias(cih_out, cih)
    c.alias(leaf_depth_out, leaf_depth)
    # circuits/main/blake3_nova.circom:6: component main {public [h, block_count, n_blocks, chunk_idx_low, chunk_idx_high]}
    outs = [n_blocks_out, block_count_out] + h_out + [total_depth_out, depth_out, cil_out, cih_out, leaf_depth_out]
    return dict(public=outs + [n_blocks, block_count] + h + [cil, cih])


# ------------------------------------------------------------------ against the reference's symbol table and witnesses
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


def lower(c, sym=None, public=()):
    """Number the signals by circom's rule, hand out the witness slots (increasing id over the non-constant classes) and
    rewrite every constraint over wires.  With the build's .sym: ids, slots and names must all be the compiler's.
    Returns (constraints over wires, nWires, wire2label, nLabels)."""
    assert len(set(c.names)) == len(c.names)
    ids = c.number()
    c.simplify(public)
    classes = {}
    for s in range(len(c.names)):
        classes.setdefault(c.find(s), []).append(s)
    # a signal that no constraint mentions and that is not part of the main component's interface has no wire
    # (Blake3GetDownLeftPath declares bit_at_depth[65] and eqs[65] but uses 64 of each: blake3_nova.circom:60-71)
    used = {c.find(k) for a, b, cc in c.cons for lc in (a, b, cc) for k in lc if k is not None} | {c.find(s) for s in public}
    used |= {c.find(s) for s, name in enumerate(c.names) if name.count(".") == 1}
    live = sorted((min(ids[s] for s in members), root) for root, members in classes.items() if root not in c.const and root in used)
    wire_of = {root: 1 + k for k, (_, root) in enumerate(live)}
    wire2label = [0] + [i for i, _ in live]
    nwires = 1 + len(live)
    if sym is not None:
        assert set(c.names) == set(sym), (len(c.names), len(sym), sorted(set(sym) - set(c.names))[:5], sorted(set(c.names) - set(sym))[:5])
        wrong = [c.names[s] for s in range(len(c.names)) if ids[s] != sym[c.names[s]][0]]
        assert not wrong, ("the numbering rule disagrees with the .sym", wrong[:5])
        for root, members in classes.items():
            kept = [sym[c.names[s]][1] for s in members if sym[c.names[s]][1] >= 0]
            if root in c.const:
                assert not kept, ("a constant signal kept a witness slot", c.names[root])
            else:
                lowest = min(members, key=lambda s: ids[s])
                assert kept == [wire_of[root]] and sym[c.names[lowest]][1] == wire_of[root], \
                    ("alias class disagrees with the .sym", [c.names[s] for s in members][:4], kept, wire_of[root])

    def low(lc):
        out = {}
        for k, v in c.resolve(lc).items():
            w = 0 if k is None else wire_of[k]
            out[w] = (out.get(w, 0) + v) % P
        return {w: f for w, f in sorted(out.items()) if f}
    cons = [(low(a), low(b), low(cc)) for a, b, cc in c.cons]
    return cons, nwires, wire2label, len(c.names) + 1


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


def summary(cons):
    kinds = {"bool": 0, "quadratic": 0, "linear": 0}
    for a, b, cc in cons:
        kinds["linear" if not a or not b else "bool" if not cc else "quadratic"] += 1
    return kinds, sum(len(a) + len(b) + len(cc) for a, b, cc in cons)


def wasm_witnesses(wasm, inputs, allow_rejects=False):
    """witnesses of the reference WASM under its own loader (tools/wasm_oracle.js), as lists of ints (None for an input the
    circuit rejects, when allowed)"""
    def strs(x):
        return [strs(y) for y in x] if isinstance(x, list) else str(x)
    with tempfile.TemporaryDirectory() as td:
        json.dump([{k: strs(v) for k, v in inp.items()} for inp in inputs], open(os.path.join(td, "in.json"), "w"))
        subprocess.check_call(["node", os.path.join(ROOT, "tools", "wasm_oracle.js"), wasm, os.path.join(td, "in.json"), os.path.join(td, "out.bin")])
        meta = json.load(open(os.path.join(td, "out.bin.err.json")))
        assert allow_rejects or not meta["errors"], meta["errors"]
        raw = open(os.path.join(td, "out.bin"), "rb").read()
    n = meta["witnessSize"]
    rejected = {int(k) for k in meta["errors"]}
    return [None if k in rejected else [int.from_bytes(raw[(k * n + i) * 32:(k * n + i + 1) * 32], "little") for i in range(n)]
            for k in range(len(inputs))]


# ------------------------------------------------------------------ the O2 builds of blake3_nova (BN254 and Vesta, 23 291 wires)
# circom's full simplification (the default, --O2) goes on after the O1 step: it eliminates one signal per linear constraint
# by substitution.  WHICH signal goes is decided inside the compiler and cannot be restated from the circuit text — but the
# reference holds both builds of the same circuit, and a kept signal keeps its value and its place in the id order.  So the
# wires of the O2 build are found by running BOTH committed WASMs on the same probe inputs: every O2 slot carries the value
# column of an O1 wire, in increasing wire order (an order-preserving alignment exists and is taken greedily; where several
# O1 wires carry the same column on every probe they are equal by the constraints themselves — bit decompositions of one
# word — or the probes below tell them apart).  The O1 wires that found no O2 slot are then eliminated from the O1 system
# through its own linear constraints (sparse Gaussian elimination over the field), which leaves the O2 build's constraint
# system up to row operations.  Nothing of this repository's kernels, slot tables or oracle enters.
def dyadic(cons):
    """Row scaling that keeps the coefficients small integers.  Eliminating a BIT b_i = (W - sum_{j != i} 2^j b_j) / 2^i puts
    2^-i into every row that mentions it — a full-width field element per term.  A * B = C holds iff (2^s A) * (2^t B) =
    2^(s+t) C: each side is multiplied by the smallest power of two that makes all its coefficients integers below 2^62 in
    magnitude (c or c - p), when there is one."""
    lim = 1 << 62

    def shift_for(lc, least=0):
        for t in range(least, 130):
            if all((f << t) % P < lim or P - (f << t) % P < lim for f in lc.values()):
                return t
        return None
    out = []
    for a, b, c in cons:
        ta, tb = shift_for(a), shift_for(b)
        if ta is None or tb is None:
            out.append((a, b, c))
            continue
        tc = shift_for(c, ta + tb)
        if tc is None:
            out.append((a, b, c))
            continue
        ta += tc - (ta + tb)                              # C needed more: put the difference on A
        if max(ta, tb, tc) == 0:
            out.append((a, b, c))
        else:
            out.append(({w: (f << ta) % P for w, f in a.items()}, {w: (f << tb) % P for w, f in b.items()}, {w: (f << tc) % P for w, f in c.items()}))
    return out


def nova_probes(seed=20260105):
    """valid nova steps that exercise every signal: random leaf / parent steps with full-range words, 64-bit chunk indices
    and large depths; for every i in [0, 64) a step with depth = total_depth - i - 2 (eqs[i] fires) for both values of bit i
    of the chunk index, as a leaf and as a parent; first / last / middle blocks; roots; message words outside [0, 2^32)
    where the circuit accepts them."""
    import random
    rng = random.Random(seed)

    def step(directed=None, parent=None, wide=False):
        n_blocks = rng.randint(1, 16)
        kind = rng.randint(0, 3)
        block_count = 0 if kind == 0 else n_blocks - 1 if kind == 1 else rng.randint(0, 40)
        depth = rng.choice([0, 0, 1, 2, 3, rng.randint(0, 60), rng.randint(0, 250)])
        par = rng.random() < 0.5 if parent is None else parent
        leaf_depth = depth + 1 if not par else depth + 2 + rng.choice([0, 0, 1, 5, rng.randint(0, 3)])
        total_depth = rng.choice([leaf_depth, depth + 2 + rng.randint(0, 63), rng.randint(0, 90), 1000])
        ci = rng.getrandbits(64)
        if directed is not None:
            i, bit = directed
            total_depth = depth + i + 2
            ci = ci & ~(1 << i) | bit << i
        word = (lambda: rng.choice([rng.getrandbits(32), -rng.randint(1, 2000), (1 << 32) + rng.getrandbits(32)])) if wide else (lambda: rng.getrandbits(32))
        return dict(n_blocks=n_blocks, block_count=block_count, h=[rng.getrandbits(32) for _ in range(8)], chunk_idx_low=ci & 0xFFFFFFFF,
                    chunk_idx_high=ci >> 32, leaf_depth=leaf_depth, total_depth=total_depth, depth=depth, m=[word() for _ in range(16)],
                    b=rng.getrandbits(32))
    probes = [step(wide=(k % 3 == 2)) for k in range(240)]
    for i in range(64):
        for bit in (0, 1):
            for par in (False, True):
                probes.append(step(directed=(i, bit), parent=par))
    return probes


def align(z_o1, z_o2):
    """O2 slot -> O1 wire: equal value columns over all probes, increasing wire order"""
    import bisect
    import collections
    K = len(z_o1)
    by_col = collections.defaultdict(list)
    for w in range(len(z_o1[0])):
        by_col[tuple(z_o1[k][w] for k in range(K))].append(w)
    amap, prev = [], -1
    for s in range(len(z_o2[0])):
        cands = by_col.get(tuple(z_o2[k][s] for k in range(K)), [])
        i = bisect.bisect_right(cands, prev)
        assert i < len(cands), f"O2 slot {s} has no O1 wire with its values after wire {prev}"
        prev = cands[i]
        amap.append(prev)
    return amap


def eliminate(cons, nwires, kept):
    """The constraint system over the kept wires only: every other wire is solved from a linear constraint and
    substituted everywhere (the constraint used up disappears).  Returns constraints over the ORIGINAL wire numbers."""
    gone = set(range(nwires)) - set(kept)
    sub_of = {}                                       # eliminated wire -> LC over wires still present

    def apply(lc):
        for _ in range(256):                           # (substitutions may mention wires that were solved later)
            if not any(w in sub_of for w in lc):
                return lc
            out = {}
            for w, f in lc.items():
                for w2, f2 in (sub_of[w].items() if w in sub_of else ((w, 1),)):
                    out[w2] = (out.get(w2, 0) + f * f2) % P
            lc = {w: f for w, f in out.items() if f}
        raise AssertionError("cyclic substitution")

    def linear_form(a, b, c):
        """the constraint as one LC = 0 if it is linear (A or B empty or a constant), else None"""
        if not a or not b:
            return neg(c)
        for x, y in ((a, b), (b, a)):
            if set(x) == {0}:
                return sub(scale(y, x[0]), c)
        return None
    rows = []
    rest = []
    for a, b, c in cons:
        lf = linear_form(a, b, c)
        (rows if lf is not None and any(w in gone for w in lf) else rest).append(lf if lf is not None and any(w in gone for w in lf) else (a, b, c))
    pending, kept_linear = list(rows), []

    def solve(lc, e):
        inv = pow(lc[e], -1, P)
        sub_of[e] = {w: (-f * inv) % P for w, f in lc.items() if w != e}
    while pending:
        changed, nxt = False, []
        for lc in pending:
            lc = apply(lc)
            unsolved = [w for w in lc if w in gone and w not in sub_of]
            if not unsolved:
                if lc:
                    kept_linear.append(lc)             # a linear constraint among kept wires: stays
            elif len(unsolved) == 1:                   # the system is nearly triangular: most rows end up here
                solve(lc, unsolved[0])
                changed = True
            else:
                nxt.append(lc)
        pending = nxt
        if pending and not changed:                    # a genuine pivot: one wire of the first open row goes
            lc = apply(pending.pop(0))
            solve(lc, next(w for w in lc if w in gone and w not in sub_of))
    for _ in range(64):                                # close the substitutions (pivot rows mention wires solved later)
        open_ = [e for e, lc in sub_of.items() if any(w in gone for w in lc)]
        if not open_:
            break
        for e in open_:
            sub_of[e] = apply(sub_of[e])
    assert not any(w in gone for lc in sub_of.values() for w in lc)
    assert not (gone - set(sub_of)), f"{len(gone - set(sub_of))} eliminated wires have no linear constraint to go through"
    pending = kept_linear
    out = []
    for lc in pending:
        lc = apply(lc)
        if lc:
            out.append(({}, {}, neg(lc)))
    for item in rest:
        a, b, c = item
        out.append((apply(a), apply(b), apply(c)))
    for a, b, c in out:
        assert not (set(a) | set(b) | set(c)) & gone
    return out


def main():
    which = sys.argv[sys.argv.index("--circuit") + 1] if "--circuit" in sys.argv else "compression"
    c = Circuit()
    if which == "compression":
        out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else OUT
        Blake3Compression(c)
        cons, nwires, w2l, nlabels = lower(c, read_sym(SYM))
        z = read_wtns(WTNS)
        assert len(z) == nwires == 24093, (len(z), nwires)
        bad = violated(cons, z)
        assert not bad, f"the reference's own witness violates {len(bad)} derived constraints, first {bad[:5]}"
        kinds, nnz = summary(cons)
        size = write_r1cs(out, cons, nwires, w2l, 16, 0, 28, nlabels)
        print(f"{len(c.names)} signals = the .sym's, ids and slots by the numbering rule = the .sym's; {nwires} wires; {len(cons)} constraints "
              f"{kinds}; {nnz} non-zeros; reference witness satisfies all; wrote {out} ({size} bytes before gzip)")
    elif which == "nova_o2":
        return main_nova_o2()
    else:
        out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else OUT.replace("blake3_compression", "blake3_nova_bn254_o1")
        nova = Blake3Nova(c)
        cons, nwires, w2l, nlabels = lower(c, public=nova["public"])
        assert nwires == 24614, nwires
        gold = json.load(open(os.path.join(ROOT, "tests", "golden", "nova_bn254_o1.json")))
        cases = [g for g in gold["cases"] if "error" not in g]
        zs = wasm_witnesses(os.path.join(REF, "build/blake3_nova/blake3_nova_js/blake3_nova.wasm"), [g["input"] for g in cases])
        for g, z in zip(cases, zs):
            bad = violated(cons, z)
            assert not bad, f"the reference WASM's witness of golden {g['name']} violates {len(bad)} derived constraints, first {bad[:5]}"
        kinds, nnz = summary(cons)
        size = write_r1cs(out, cons, nwires, w2l, 15, 12, 20, nlabels)
        print(f"{len(c.names)} signals; {nwires} wires = the WASM's witness size; {len(cons)} constraints {kinds}; {nnz} non-zeros; "
              f"{len(cases)} reference-WASM witnesses (accepted goldens) satisfy all; wrote {out} ({size} bytes before gzip)")


VESTA_Q = 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001      # circom's "vesta" prime


def main_nova_o2():
    """blake3_nova_bn254.r1cs.gz and blake3_nova_vesta.r1cs.gz (the two O2 builds, 23 291 wires)"""
    global P
    outdir = sys.argv[sys.argv.index("--outdir") + 1] if "--outdir" in sys.argv else os.path.dirname(OUT)
    wasm = {"o1": "build/blake3_nova/blake3_nova_js/blake3_nova.wasm", "bn254": "build/blake3_nova_js/blake3_nova.wasm",
            "vesta": "build/blake3_nova_pasta_js/blake3_nova_pasta.wasm"}
    probes = nova_probes()
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "nova_bn254.json")))
    probes += [g["input"] for g in gold["cases"] if "error" not in g]
    z = {k: wasm_witnesses(os.path.join(REF, v), probes, allow_rejects=True) for k, v in wasm.items()}
    ok = [k for k in range(len(probes)) if all(z[b][k] is not None for b in z)]
    assert len(ok) >= 400, len(ok)
    # the O1 system (BN254) and the alignment of the O2 slots to its wires
    c = Circuit()
    nova = Blake3Nova(c)
    cons1, nw1, w2l1, nlabels = lower(c, public=nova["public"])
    assert nw1 == 24614
    for k in ok:
        assert not violated(cons1, z["o1"][k])
    amap = align([z["o1"][k] for k in ok], [z["bn254"][k] for k in ok])
    assert amap[:46] == list(range(46)) and len(amap) == 23291, amap[:50]
    slot_of = {w: s_ for s_, w in enumerate(amap)}
    for name, prime in (("bn254", P), ("vesta", VESTA_Q)):
        P = prime
        c = Circuit()
        nova = Blake3Nova(c)
        cons_o1, nw, w2l, _ = lower(c, public=nova["public"])          # the same circuit over this build's field
        assert nw == nw1 and w2l == w2l1
        cons = dyadic(eliminate(cons_o1, nw, amap))
        remap = lambda lc: {slot_of[w]: f for w, f in sorted(lc.items(), key=lambda kv: slot_of[kv[0]])}
        cons = [(remap(a), remap(b), remap(cc)) for a, b, cc in cons]
        bad = [(k, violated(cons, z[name][k])[:3]) for k in ok if violated(cons, z[name][k])]
        assert not bad, f"{name}: reference-WASM witnesses violate the derived system: {bad[:3]}"
        kinds, nnz = summary(cons)
        out = os.path.join(outdir, f"blake3_nova_{name}.r1cs.gz")
        size = write_r1cs(out, cons, len(amap), [w2l1[w] for w in amap], 15, 12, 20, nlabels)
        print(f"nova {name} (O2): {len(amap)} wires aligned to the circomkit build's {nw1}; {nw1 - len(amap)} wires eliminated through linear "
              f"constraints; {len(cons)} constraints {kinds}; {nnz} non-zeros; {len(ok)} reference-WASM witnesses satisfy all; wrote {out} "
              f"({size} bytes before gzip)")
    P = 21888242871839275222246405745257275088548364400416034343698204186575808495617


if __name__ == "__main__":
    sys.exit(main())
