"""Synthetic commitment keys for benchmarks and examples: `n` deterministic points on the circuit's curve.

A real deployment passes arecibo's commitment key (rust_fold/src/main.rs:205-258 derives it from the circuit
shape); there is no such key in the reference checkout, so the bench makes one by try-and-increment on a
SHA-256 stream: x = H(seed, i, ctr) mod p until x^3 + b is a square.  Plain Python integers; the result is
what CommitKey(ctx, curve, generators) takes: 64 bytes per point, x then y, 32-byte little-endian each.
"""
import hashlib

# y^2 = x^3 + b over the base field p
CURVES = {
    "bn254_g1": (21888242871839275222246405745257275088696311157297823662689037894645226208583, 3),
    "vesta": (0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001, 5),     # the Pallas curve (scalar field = circom's "vesta" prime)
}
CURVES["pallas"] = CURVES["vesta"]


def _sqrt(a, p):
    """Square root mod p (p = 3 mod 4 directly, otherwise Tonelli-Shanks), None for a non-residue."""
    a %= p
    if a == 0:
        return 0
    if pow(a, (p - 1) >> 1, p) != 1:
        return None
    if p & 3 == 3:
        return pow(a, (p + 1) >> 2, p)
    q, s = p - 1, 0
    while not q & 1:
        q >>= 1
        s += 1
    z = 2
    while pow(z, (p - 1) >> 1, p) != p - 1:
        z += 1
    c, t, r, m = pow(z, q, p), pow(a, q, p), pow(a, (q + 1) >> 1, p), s
    while t != 1:
        i, u = 0, t
        while u != 1:
            u = u * u % p
            i += 1
        e = pow(c, 1 << (m - i - 1), p)
        r, c = r * e % p, e * e % p
        t, m = t * c % p, i
    return r


def generators(curve, n, seed=b"b3wit-synthetic-key"):
    """n * 64 bytes: affine points (x, y) of `curve`, little-endian, standard (non-Montgomery) form."""
    p, b = CURVES[curve]
    out = bytearray()
    for i in range(n):
        ctr = 0
        while True:
            h = hashlib.sha256(seed + i.to_bytes(4, "little") + ctr.to_bytes(4, "little")).digest()
            x = int.from_bytes(h + hashlib.sha256(h).digest()[:8], "little") % p
            y = _sqrt(x * x * x + b, p)
            if y:
                if (y ^ h[0]) & 1:
                    y = p - y
                out += x.to_bytes(32, "little") + y.to_bytes(32, "little")
                break
            ctr += 1
    return bytes(out)
