"""tests/ec_ref.py (the plain-integer group law the commitment kernel is checked against) pinned to PUBLISHED constants,
since the reference holds no commitment vectors (arecibo 0.1.1 is a crates.io dependency, rust_fold/Cargo.lock:75):

* alt_bn128 / BN254 G1 (EIP-196): generator (1, 2); 2G and 3G as printed in EIP-196 / py_ecc; the bn256Add and
  bn256ScalarMul precompile known answers "chfast1" of go-ethereum's core/vm/testdata/precompiles; group order = the
  circuits' BN254 prime r (r * G = infinity).
* Pasta (pasta_curves): both curves are y^2 = x^3 + 5 with generator (-1, 2); Pallas lives over
  p = 0x4000...094cf91b992d30ed00000001 and has order q = 0x4000...0994a8dd8c46eb2100000001 — the prime of the reference's
  `--prime vesta` build (circom calls the field "vesta"; the group whose SCALAR field it is, i.e. the one arecibo's
  PallasEngine commits in, rust_fold/src/main.rs:366, is Pallas).  ec_ref / the library call that curve "vesta" after the
  circuit's prime name.
So: group law pinned to public vectors; arecibo's commitment-KEY derivation stays unpinned."""
import ec_ref as E
import b3w_testlib as T

H = lambda s: int(s, 16)
Q_BN = 21888242871839275222246405745257275088696311157297823662689037894645226208583
P_PALLAS = 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001


def test_curve_constants():
    assert E.CURVES["bn254_g1"] == (Q_BN, 3)
    assert E.CURVES["vesta"] == (P_PALLAS, 5)
    assert T.VESTA_Q == 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001


def test_bn254_generator_multiples_eip196():
    G = (1, 2)
    assert E.on_curve(G, "bn254_g1")
    G2 = (H("030644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd3"), H("15ed738c0e0a7c92e7845f96b2ae9c0a68a6a449e3538fc7ff3ebf7a5a18a2c4"))
    G3 = (3353031288059533942658390886683067124040920775575537747144343083137631628272,
          19321533766552368860946552437480515441416830039777911637913418824951667761761)
    assert E.add(G, G, Q_BN) == G2 == E.mul(2, G, Q_BN)
    assert E.add(G2, G, Q_BN) == G3 == E.mul(3, G, Q_BN)
    assert E.mul(T.BN254_R, G, Q_BN) is None and E.mul(T.BN254_R - 1, G, Q_BN) == E.neg(G, Q_BN)     # prime order r
    assert E.add(G, E.neg(G, Q_BN), Q_BN) is None and E.add(None, G, Q_BN) == G


def test_bn256_precompile_known_answers():
    # bn256Add "chfast1"
    P1 = (H("18b18acfb4c2c30276db5411368e7185b311dd124691610c5d3b74034e093dc9"), H("063c909c4720840cb5134cb9f59fa749755796819658d32efc0d288198f37266"))
    P2 = (H("07c2b7f58a84bd6145f00c9c2bc0bb1a187f20ff2c92963a88019e7c6a014eed"), H("06614e20c147e940f2d70da3f74c9a17df361706a4485c742bd6788478fa17d7"))
    R = (H("2243525c5efd4b9c3d3c45ac0ca3fe4dd85e830a4ce6b65fa1eeaee202839703"), H("301d1d33be6da8e509df21cc35964723180eed7532537db9ae5e7d48f195c915"))
    assert all(E.on_curve(X, "bn254_g1") for X in (P1, P2, R)) and E.add(P1, P2, Q_BN) == R
    # bn256ScalarMul "chfast1"
    Q = (H("2bd3e6d0f3b142924f5ca7b49ce5b9d54c4703d7ae5648e61d02268b1a0a9fb7"), H("21611ce0a6af85915e2f1d70300909ce2e49dfad4a4619c8390cae66cefdb204"))
    k = H("11138ce750fa15c2")
    S = (H("070a8d6a982153cae4be29d434e8faef8a47b274a053f5a4ee2a6c9c13c31e5c"), H("031b8ce914eba3a9ffb989f9cdd5b0f01943074bf4f0f315690ec3cec6981afc"))
    assert E.mul(k, Q, Q_BN) == S
    assert E.commit([k, 1, 0, 2], [Q, P1, P2, P2], "bn254_g1") == E.add(E.add(S, P1, Q_BN), E.add(P2, P2, Q_BN), Q_BN)


def test_pallas_generator_and_order():
    G = (P_PALLAS - 1, 2)                                   # (-1, 2)
    assert E.on_curve(G, "vesta")
    assert E.mul(T.VESTA_Q, G, P_PALLAS) is None            # the group's order is the circuit's ("vesta") prime
    assert E.mul(T.VESTA_Q - 1, G, P_PALLAS) == E.neg(G, P_PALLAS)
    assert E.mul(T.VESTA_Q + 5, G, P_PALLAS) == E.mul(5, G, P_PALLAS)
    # sqrt_mod on a 2-adic field (p - 1 = 2^32 * odd): Tonelli-Shanks
    y = E.sqrt_mod(4, P_PALLAS)
    assert y in (2, P_PALLAS - 2)


def test_synthetic_key_points_are_on_the_curves():
    import importlib
    K = importlib.import_module("hot-proofs-blake3-circom_amd.synthetic_key")
    assert K.CURVES == E.CURVES and E.CURVES["pallas"] == E.CURVES["vesta"]
    for curve in ("bn254_g1", "pallas"):
        g = K.generators(curve, 40, seed=b"t")
        pts = [E.point_from_bytes(g[64 * i:64 * i + 64]) for i in range(40)]
        assert all(E.on_curve(P, curve) and P is not None for P in pts) and len(set(pts)) == 40
