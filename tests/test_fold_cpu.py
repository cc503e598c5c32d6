"""fold.py on the CPU: the .r1cs parser sums a wire that is repeated inside one A / B / C list, as the device
parser does (csrc/b3w_r1cs_host.cpp keeps both terms), and generator folding refuses a system whose prime is not the order of the
curve it is asked to fold on — the relations w_k = sum a_kj w_j only carry over to the points when it is."""
import importlib
import numpy as np
import pytest
import b3w_testlib as T
import r1cs_ref as R
import ec_ref as E

fold = importlib.import_module("hot-proofs-blake3-circom_amd.fold")


def test_repeated_wires_add_up():
    p = T.BN254_R
    # 0 * 0 = w1 + w1 - 2 w2   (wire 1 twice in C) ; (w3 + w3) * 1 = w2, wire 3 twice in A; a pair that cancels
    cons = [([], [], [(1, 1), (1, 1), (2, p - 2)]), ([(3, 1), (3, 1)], [(0, 1)], [(2, 1)]), ([(4, 5), (4, p - 5)], [(0, 1)], [])]
    prime, nw, got = fold.parse_r1cs(R.write_image(p, 5, cons))
    assert prime == p and nw == 5
    assert got[0] == ({}, {}, {1: 2, 2: p - 2})
    assert got[1] == ({3: 2}, {0: 1}, {2: 1})
    assert got[2] == ({}, {0: 1}, {})                       # the cancelling pair is gone, not a zero term
    rel = fold.linear_relations(prime, got)
    assert {1: 2, 2: p - 2} in rel


def test_folding_needs_the_curves_own_scalar_field():
    img = R.write_image(T.BN254_R, 3, [([], [], [(1, 1), (2, T.BN254_R - 1)])])
    gens = E.points_to_bytes(E.random_points("vesta", 3, seed=b"x"))
    with pytest.raises(ValueError, match="not the order of"):
        fold.fold_generators(img, [1, 32, 32], 0, gens, "pallas")
    gens = E.points_to_bytes(E.random_points("bn254_g1", 3, seed=b"x"))
    out, mask, stats = fold.fold_generators(img, [1, 32, 32], 0, gens, "bn254_g1")        # w1 = w2: one of them folds into the other
    assert sum(1 for m in mask if m) == 1
