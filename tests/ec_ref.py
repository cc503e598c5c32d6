"""Short-Weierstrass curves with a = 0 over a prime field in plain Python integers — an independent checker for the
commitment kernel (csrc/b3w_commit.hip).  Affine formulas straight from the group law; None = the point at infinity.
Test infrastructure."""
import hashlib

CURVES = {
    # BN254 / alt_bn128 G1: y^2 = x^3 + 3 over q
    "bn254_g1": (21888242871839275222246405745257275088696311157297823662689037894645226208583, 3),
    # Pallas (pasta_curves): y^2 = x^3 + 5 over p below; its scalar field is the prime circom calls "vesta", which is why the
    # key is "vesta" here and in the library's older name B3W_CURVE_VESTA (tests/test_ec_ref_public_vectors.py pins it)
    "vesta": (0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001, 5),
}
CURVES["pallas"] = CURVES["vesta"]
CURVE_ID = {"bn254_g1": 0, "vesta": 1, "pallas": 1}


def sqrt_mod(a, p):
    """Tonelli-Shanks; None if a is not a square."""
    a %= p
    if a == 0:
        return 0
    if pow(a, (p - 1) // 2, p) != 1:
        return None
    if p % 4 == 3:
        return pow(a, (p + 1) // 4, p)
    q, s = p - 1, 0
    while q % 2 == 0:
        q //= 2
        s += 1
    z = 2
    while pow(z, (p - 1) // 2, p) != p - 1:
        z += 1
    m, c, t, r = s, pow(z, q, p), pow(a, q, p), pow(a, (q + 1) // 2, p)
    while t != 1:
        i, t2 = 0, t
        while t2 != 1:
            t2 = t2 * t2 % p
            i += 1
        b = pow(c, 1 << (m - i - 1), p)
        m, c = i, b * b % p
        t, r = t * c % p, r * b % p
    return r


def add(P, Q, p):
    if P is None:
        return Q
    if Q is None:
        return P
    (x1, y1), (x2, y2) = P, Q
    if x1 == x2:
        if (y1 + y2) % p == 0:
            return None
        lam = 3 * x1 * x1 * pow(2 * y1, -1, p) % p
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, p) % p
    x3 = (lam * lam - x1 - x2) % p
    return x3, (lam * (x1 - x3) - y1) % p


def mul(k, P, p):
    R = None
    while k:
        if k & 1:
            R = add(R, P, p)
        P = add(P, P, p)
        k >>= 1
    return R


def neg(P, p):
    return None if P is None else (P[0], (-P[1]) % p)


_POINTS = {}                                                   # (curve, seed) -> the points made so far: point i depends on (seed, i) only


def random_points(curve, n, seed=b"b3wit-test-generators"):
    """n points by try-and-increment on a hash of (seed, i, counter).  (Kept per (curve, seed) for the process: a witness's worth of
    points is 5-10 s of square roots in Python, and the GPU suite asks for the same ones again and again.)"""
    p, b = CURVES[curve]
    out = _POINTS.setdefault((curve, bytes(seed)), [])
    for i in range(len(out), n):
        ctr = 0
        while True:
            h = hashlib.sha256(seed + i.to_bytes(4, "little") + ctr.to_bytes(4, "little")).digest()
            x = int.from_bytes(h + hashlib.sha256(h).digest()[:8], "little") % p
            y = sqrt_mod(x * x * x + b, p)
            if y is not None and y != 0:
                out.append((x, y if h[0] & 1 else p - y))
                break
            ctr += 1
    return list(out[:n])


def on_curve(P, curve):
    p, b = CURVES[curve]
    return P is None or (P[1] * P[1] - P[0] ** 3 - b) % p == 0


def commit(values, gens, curve):
    """sum values[i] * gens[i]"""
    p, _ = CURVES[curve]
    acc = None
    for v, G in zip(values, gens):
        if v == 0:
            continue
        acc = add(acc, G if v == 1 else mul(v, G, p), p)
    return acc


def points_to_bytes(pts):
    return b"".join((b"\0" * 64) if P is None else P[0].to_bytes(32, "little") + P[1].to_bytes(32, "little") for P in pts)


def point_from_bytes(b):
    x, y = int.from_bytes(b[:32], "little"), int.from_bytes(b[32:64], "little")
    return None if x == 0 and y == 0 else (x, y)
