"""Host-side index math of the chained-mode planner (no GPU needed): path lengths, parent-step rows and the "provable"
predicate for arbitrary chunk counts, against a direct walk of BLAKE3's tree and against what the reference WASM did
(tests/golden/incomplete_trees.nova_vesta.json.gz, tools/probe_incomplete_trees.js)."""
import gzip
import json
import os

import b3w_testlib as T


def _walk(c, n):
    """true directions of chunk c's path, root first (True = left); BLAKE3: left subtree = largest power of two below n"""
    dirs = []
    while n > 1:
        k = 1
        while k * 2 < n:
            k *= 2
        if c < k:
            n = k
            dirs.append(True)
        else:
            c, n = c - k, n - k
            dirs.append(False)
    return dirs


def test_rows_lengths_and_provable_for_every_count_up_to_300():
    L = T.pkg().lib()
    for n in list(range(1, 131)) + [255, 256, 257, 300]:
        row = 0
        for c in range(n):
            dirs = _walk(c, n)
            p = len(dirs)
            assert L.b3w_chain_path_len(c, n) == p
            assert L.b3w_chain_parent_row(c, n) == row, (n, c)
            want = all((((c >> (p - 1 - i)) & 1) == 0) == d for i, d in enumerate(dirs))
            assert bool(L.b3w_chain_path_provable(c, n)) == want, (n, c)
            row += p
        assert L.b3w_chain_parent_row(n, n) == row
        assert L.b3w_chain_num_parent_steps(n * 1024, 0, n) == row
        if n > 3:
            assert L.b3w_chain_num_parent_steps(n * 1024, 1, n - 2) == row - len(_walk(0, n)) - len(_walk(n - 1, n))
        if n & (n - 1) == 0:
            assert all(L.b3w_chain_path_provable(c, n) for c in range(n))
    assert L.b3w_chain_path_provable(5, 5) == 0 and L.b3w_chain_num_parent_steps(4096, 3, 2) == 0


def test_provable_is_what_the_reference_wasm_did():
    doc = json.load(gzip.open(os.path.join(T.GOLD, "incomplete_trees.nova_vesta.json.gz"), "rt"))
    L = T.pkg().lib()
    seen = 0
    for tree in doc["trees"]:
        n = tree["n_chunks"]
        for leaf in tree["leaves"]:
            assert leaf["error"] is None
            assert L.b3w_chain_path_len(leaf["leaf"], n) == leaf["path_len"]
            assert bool(L.b3w_chain_path_provable(leaf["leaf"], n)) == leaf["ends_in_root"] == leaf["bits_agree"]
            assert (leaf["final_h"] == tree["root"]) == leaf["ends_in_root"]
            seen += 1
    assert seen >= 100 and {t["n_chunks"] for t in doc["trees"]} >= {3, 5, 6, 7, 100}
