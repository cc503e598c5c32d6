"""The vectorised numpy BLAKE3 (tests/blake3_ref.py: chunk_cvs_np / tree_levels_np, the checker of the 1 GiB chained
pass) against the pure-Python one written from the specification, and that one against BLAKE3's published test vector."""
import numpy as np

import b3w_testlib as T
import blake3_ref as B


def test_numpy_tree_equals_the_scalar_hash():
    W = T.workloads()
    for nchunks in (2, 8, 64):
        data = W.lcg_preimage(nchunks * 1024, seed=1)
        cvs = B.chunk_cvs_np(data)
        for c in (0, nchunks - 1):
            assert list(cvs[c]) == B.chunk_cv(data[c * 1024:(c + 1) * 1024].tobytes(), c, False)
        levels = B.tree_levels_np(cvs)
        assert len(levels) == nchunks.bit_length() and list(levels[-1][0]) == B.hash_words(data.tobytes())


def test_scalar_hash_known_answers():
    # BLAKE3 official test vectors (test_vectors.json: input byte i = i % 251): empty input and 1 byte
    assert B.blake3(b"").hex() == "af1349b9f5f9a1a6a0404dea36dcc9499bcb25c9adc112b7cc9a93cae41f3262"
    assert B.blake3(bytes([0])).hex() == "2d3adedff11b61f14c886e35afa036736dcd87a74d27b5c1510225d0f592e213"
    assert B.blake3(bytes(i % 251 for i in range(1025))).hex() == "d00278ae47eb27b34faecf67b4fe263f82d5412916c1ffd97c8cb7fb814b8444"
