"""fold.py — generator folding for commitment keys (include/b3wit.h "FOLDED keys").

A witness satisfies its circuit's linear constraints.  Where such a constraint expresses a wide slot through others — in these
circuits every 32-bit word is the sum of its bit slots — the word need not be committed by itself:

    w_k = sum_j a_kj w_j      =>      w_k G_k = sum_j w_j (a_kj G_k)

so a_kj G_k is added to the generator of slot j once, and slot k drops out of the multi-scalar multiplication.  This module
derives the relations from an iden3 .r1cs image (Gaussian elimination over the rows that are linear: an empty A or B side, or one
that is a constant), keeps only eliminations whose expression stays inside the committed slots, and folds the caller's generators
(plain-integer affine arithmetic on the curve; a second or so per key).  The point committed for a witness does not change;
tests/test_gpu_commit.py compares folded and unfolded keys on the device.

What arecibo does with the key is not touched: the caller still hands in arecibo's generators, slot by slot.
"""
import struct

from .synthetic_key import CURVES

# order of each curve's group = the field the circuit must be over for its linear relations to carry over to the points
GROUP_ORDER = {"bn254_g1": 21888242871839275222246405745257275088548364400416034343698204186575808495617,
               "pallas": 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001}
GROUP_ORDER["vesta"] = GROUP_ORDER["pallas"]              # (older name of the same curve id, after the circuit's prime)


def parse_r1cs(img):
    """-> (prime, n_wires, [(A, B, C)]) with A, B, C = {wire: coefficient}"""
    if img[:4] != b"r1cs" or struct.unpack_from("<I", img, 4)[0] != 1:
        raise ValueError("not an r1cs v1 image")
    nsec = struct.unpack_from("<I", img, 8)[0]
    pos, secs = 12, {}
    for _ in range(nsec):
        typ, size = struct.unpack_from("<IQ", img, pos)
        secs[typ] = (pos + 12, size)
        pos += 12 + size
    h, _ = secs[1]
    fs = struct.unpack_from("<I", img, h)[0]
    prime = int.from_bytes(img[h + 4:h + 4 + fs], "little")
    nw, _, _, _, _, m = struct.unpack_from("<IIIIQI", img, h + 4 + fs)
    c, _ = secs[2]
    cons = []
    for _ in range(m):
        parts = []
        for _ in range(3):
            n = struct.unpack_from("<I", img, c)[0]
            c += 4
            lc = {}
            for _ in range(n):
                w = struct.unpack_from("<I", img, c)[0]
                # a wire repeated inside one list ADDS its terms (the device parser, b3w_r1cs_host.cpp, keeps both)
                lc[w] = (lc.get(w, 0) + int.from_bytes(img[c + 4:c + 4 + fs], "little")) % prime
                c += 4 + fs
            parts.append({w: cf for w, cf in lc.items() if cf})
        cons.append(tuple(parts))
    return prime, nw, cons


def linear_relations(prime, cons):
    """the rows that say  sum_j c_j w_j = 0"""
    p = prime
    out = []
    for a, b, c in cons:
        if not a or not b:                                  # 0 * B = C
            out.append(dict(c))
        elif set(a) == {0} or set(b) == {0}:                # (k * 1) * B = C, wire 0 being the constant 1
            k, other = (a[0], b) if set(a) == {0} else (b[0], a)
            row = {w: (-k * cf) % p for w, cf in other.items()}
            for w, cf in c.items():
                row[w] = (row.get(w, 0) + cf) % p
            out.append({w: cf for w, cf in row.items() if cf})
    return out


def bit_of_word_relations(prime, n_wires, cons, widths):
    """What circom's O2 pass leaves of a word's decomposition  W = sum 2^j b_j  after it has used the linear constraint to take bit
    i out of the witness: the booleanity of that bit,  N (N - 2^i) = 0  with  N = W - sum_{j != i} 2^j b_j  (up to scaling, the
    constant on either side).  Then  W = sum_{j != i} 2^j b_j + 2^i beta  with beta = bit i of W's value.  Returned as linear rows
    over the wires plus one VIRTUAL wire n_wires + W per word (its beta), and {W: i}.  Only the clean case is taken: one 32-bit
    word, 31 distinct bit slots with coefficients 2^j, roots 0 and 2^i, all 32 positions covered."""
    p = prime
    rows, bit_of = [], {}
    for a, b, c in cons:
        if c:
            continue
        wa, wb = set(a) - {0}, set(b) - {0}
        if wa != wb or len(wa) != 32:
            continue
        words = [w for w in wa if widths[w] != 1]
        if len(words) != 1 or widths[words[0]] != 32 or words[0] in bit_of:
            continue
        W = words[0]
        ia, ib = pow(a[W], -1, p), pow(b[W], -1, p)
        na = {w: cf * ia % p for w, cf in a.items()}
        nb = {w: cf * ib % p for w, cf in b.items()}
        if any(na[w] != nb[w] for w in wa):
            continue
        roots = sorted([(-na.get(0, 0)) % p, (-nb.get(0, 0)) % p])      # N = W + sum c_j b_j is one of these
        d = roots[1]
        if roots[0] != 0 or d == 0 or d & (d - 1) or d.bit_length() > 32:
            continue
        i = d.bit_length() - 1
        J = {}
        for w in wa:
            if w == W:
                continue
            cj = (-na[w]) % p                                            # W = sum cj b_j + N
            if cj == 0 or cj & (cj - 1) or cj.bit_length() > 32 or (cj.bit_length() - 1) in J:
                J = None
                break
            J[cj.bit_length() - 1] = w
        if J is None or i in J or set(J) | {i} != set(range(32)):
            continue
        row = {W: 1, n_wires + W: (-d) % p}
        for j, w in J.items():
            row[w] = (-(1 << j)) % p
        rows.append(row)
        bit_of[W] = i
    return rows, bit_of


def eliminate(prime, n_wires, rows, widths, first_slot):
    """-> {k: {j: a_kj}}: slot k (wider than one bit) as a combination of kept slots, all of them committed (>= first_slot)."""
    p = prime
    expr = {}

    def substitute(row):
        again = True
        while again:
            again = False
            for w in list(row):
                if w in expr:
                    cf = row.pop(w)
                    for w2, c2 in expr[w].items():
                        v = (row.get(w2, 0) + cf * c2) % p
                        if v:
                            row[w2] = v
                        else:
                            row.pop(w2, None)
                    again = True
        return row

    for row in rows:
        row = substitute(dict(row))
        cands = [w for w, cf in row.items() if first_slot <= w < n_wires and widths[w] > 1 and cf in (1, p - 1)]
        if not cands:
            continue
        k = max(cands, key=lambda w: (widths[w], w))
        ck = row.pop(k)
        inv = pow(ck, -1, p)
        e = {w: (-cf * inv) % p for w, cf in row.items()}
        expr[k] = e
        for k2, e2 in expr.items():                         # back-substitute into what was eliminated before
            if k2 != k and k in e2:
                cf = e2.pop(k)
                for w2, c2 in e.items():
                    v = (e2.get(w2, 0) + cf * c2) % p
                    if v:
                        e2[w2] = v
                    else:
                        e2.pop(w2, None)
    # only what stays inside the committed range, over slots that are kept
    return {k: e for k, e in expr.items() if all(w >= first_slot and w not in expr for w in e)}   # (virtual wires: >= n_wires)


class _Curve:
    """y^2 = x^3 + b over F_q, affine, None = infinity (plain integers: set-up only)"""

    def __init__(self, q, b):
        self.q, self.b = q, b

    def add(self, P, Q):
        q = self.q
        if P is None:
            return Q
        if Q is None:
            return P
        x1, y1 = P
        x2, y2 = Q
        if x1 == x2:
            if (y1 + y2) % q == 0:
                return None
            lam = 3 * x1 * x1 * pow(2 * y1, -1, q) % q
        else:
            lam = (y2 - y1) * pow(x2 - x1, -1, q) % q
        x3 = (lam * lam - x1 - x2) % q
        return x3, (lam * (x1 - x3) - y1) % q

    def neg(self, P):
        return None if P is None else (P[0], (-P[1]) % self.q)

    def mul(self, k, P):
        R = None
        while k:
            if k & 1:
                R = self.add(R, P)
            P = self.add(P, P)
            k >>= 1
        return R


def fold_generators(image, widths, first_slot, generators, curve):
    """-> (folded generators: bytes like `generators`, mask: bytearray (1 = slot folded away), stats dict)"""
    prime, n_wires, cons = parse_r1cs(image)
    if n_wires != len(widths):
        raise ValueError("the constraint system is not this circuit's")
    if prime != GROUP_ORDER[curve]:
        raise ValueError(f"the constraint system's prime is not the order of {curve}: its linear relations do not carry over to the points")
    widths = [int(w) for w in widths]
    bow_rows, bit_of = bit_of_word_relations(prime, n_wires, cons, widths)
    expr = eliminate(prime, n_wires, linear_relations(prime, cons) + bow_rows, widths, first_slot)
    # a word whose relation was not used for ITS elimination keeps its slot as it is (its virtual wire then stands for nothing)
    bit_of = {W: i for W, i in bit_of.items() if W in expr and (n_wires + W) in expr[W]}
    if any(w >= n_wires and (w - n_wires) not in bit_of for e in expr.values() for w in e):
        raise ValueError("a virtual bit is used outside its word's elimination")
    q, b = CURVES[curve]
    E = _Curve(q, b)
    nslots = n_wires - first_slot
    G = []
    for i in range(nslots):
        x = int.from_bytes(generators[64 * i:64 * i + 32], "little")
        y = int.from_bytes(generators[64 * i + 32:64 * i + 64], "little")
        G.append(None if x == 0 and y == 0 else (x, y))
    out = list(G)
    virt = {}                                               # generators of the virtual bits: they start at infinity

    def acc(j, T):
        if j >= n_wires:
            virt[j - n_wires] = E.add(virt.get(j - n_wires), T)
        else:
            out[j - first_slot] = E.add(out[j - first_slot], T)
    half = prime >> 1
    for k, e in expr.items():
        Gk = G[k - first_slot]
        # multiples of G_k by doubling, shared by the coefficients that are +-2^i (all of them in these circuits)
        pow2 = {0: Gk}
        top = 0
        for j, a in e.items():
            neg = a > half
            mag = prime - a if neg else a
            if mag & (mag - 1) == 0:
                i = mag.bit_length() - 1
                while top < i:
                    pow2[top + 1] = E.add(pow2[top], pow2[top])
                    top += 1
                T = pow2[i]
            else:
                T = E.mul(mag, Gk)
            acc(j, E.neg(T) if neg else T)
    mask = bytearray(nslots)
    buf = bytearray(generators)
    for k in expr:
        mask[k - first_slot] = 1
    for W, i in bit_of.items():                             # the word's slot now stands for bit i of its value
        mask[W - first_slot] = 0x80 | i
        out[W - first_slot] = virt[W]
    for i, P in enumerate(out):
        if P is None:
            if mask[i] != 1:
                raise ValueError("a folded generator is the point at infinity")     # (cannot happen with independent generators)
            continue
        buf[64 * i:64 * i + 32] = P[0].to_bytes(32, "little")
        buf[64 * i + 32:64 * i + 64] = P[1].to_bytes(32, "little")
    v0 = sum(widths[first_slot:])
    v1 = sum((1 if mask[i] & 0x80 else w) for i, w in enumerate(widths[first_slot:]) if mask[i] != 1)
    return bytes(buf), mask, {"folded_slots": len(expr), "single_bit_words": len(bit_of), "virtual_slots": v0, "virtual_slots_folded": v1,
                              "terms": sum(len(e) for e in expr.values())}
