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
                lc[w] = int.from_bytes(img[c + 4:c + 4 + fs], "little")
                c += 4 + fs
            parts.append(lc)
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
        cands = [w for w, cf in row.items() if w >= first_slot and widths[w] > 1 and cf in (1, p - 1)]
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
    return {k: e for k, e in expr.items() if all(w >= first_slot and w not in expr for w in e)}


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
    widths = [int(w) for w in widths]
    expr = eliminate(prime, n_wires, linear_relations(prime, cons), widths, first_slot)
    q, b = CURVES[curve]
    E = _Curve(q, b)
    nslots = n_wires - first_slot
    G = []
    for i in range(nslots):
        x = int.from_bytes(generators[64 * i:64 * i + 32], "little")
        y = int.from_bytes(generators[64 * i + 32:64 * i + 64], "little")
        G.append(None if x == 0 and y == 0 else (x, y))
    out = list(G)
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
            out[j - first_slot] = E.add(out[j - first_slot], E.neg(T) if neg else T)
    mask = bytearray(nslots)
    buf = bytearray(generators)
    for k in expr:
        mask[k - first_slot] = 1
    for i, P in enumerate(out):
        if P is None:
            if not mask[i]:
                raise ValueError("a folded generator is the point at infinity")     # (cannot happen with independent generators)
            continue
        buf[64 * i:64 * i + 32] = P[0].to_bytes(32, "little")
        buf[64 * i + 32:64 * i + 64] = P[1].to_bytes(32, "little")
    v0 = sum(widths[first_slot:])
    v1 = sum(w for i, w in enumerate(widths[first_slot:]) if not mask[i])
    return bytes(buf), mask, {"folded_slots": len(expr), "virtual_slots": v0, "virtual_slots_folded": v1,
                              "terms": sum(len(e) for e in expr.values())}
