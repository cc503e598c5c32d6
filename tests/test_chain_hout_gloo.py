"""The fold's exchange in chained mode on CPU: world_size-2 gloo.  Each rank holds the public outputs of its contiguous
chunk shard (here made with the plain BLAKE3 of tests/blake3_ref.py standing in for the device pass, which needs a GPU)
and chain.gather_h_out all-gathers h_out (public words 2..9) of every step into global step order: row 16 c + blocks(c) - 1
of the gathered array is chunk c's chaining value on both ranks (z_{i+1} = outputs of step i,
rust_fold/src/blake3_circuit.rs:111-123; BASELINE config 4 "RCCL gather of h_out")."""
import os
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import b3w_testlib as T
import blake3_ref as B


def _leaf_h_out(data, c):
    """running chaining value after every block of chunk c: what the step circuit outputs as h_out (a preimage of one chunk
    is its own root: ROOT on its last block, Blake3GetFlag, circuits/blake3_nova.circom:122-167)"""
    chunk = data[c * 1024:(c + 1) * 1024]
    blocks = [chunk[i:i + 64] for i in range(0, len(chunk), 64)] or [b""]
    cv, rows = B.IV, []
    for i, blk in enumerate(blocks):
        d = (B.CHUNK_START if i == 0 else 0) | (B.CHUNK_END if i == len(blocks) - 1 else 0)
        if len(data) <= 1024 and i == len(blocks) - 1:
            d |= B.ROOT
        cv = B.compress(cv, B._words(blk), c, len(blk), d)[:8]
        rows.append(cv)
    return rows


def _worker(rank, world, port, nbytes, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        chain = __import__("importlib").import_module("hot-proofs-blake3-circom_amd.chain")
        data = T.workloads().lcg_preimage(nbytes, seed=1).tobytes()
        f, k, n_leaf, p0, n_par = chain.step_shards(nbytes, world)[rank]
        pub = np.full((n_leaf + n_par, 15), 0xDEAD0000 + rank, dtype=np.uint32)          # words outside 2..9 never travel
        rows = [r for c in range(f, f + k) for r in _leaf_h_out(data, c)]
        assert len(rows) == n_leaf
        if n_leaf:
            pub[:n_leaf, 2:10] = np.array(rows, dtype=np.uint32)
        pub[n_leaf:, 2:10] = (np.arange(p0, p0 + n_par, dtype=np.uint32)[:, None] * 8 + np.arange(8, dtype=np.uint32)[None, :])
        leaf, par = chain.gather_h_out(torch.from_numpy(pub.view(np.int32)), n_leaf, nbytes)
        ret[rank] = (leaf.numpy().view(np.uint32).copy(), par.numpy().view(np.uint32).copy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("nbytes", [64 * 1024, 3 * 1024 + 5, 700, 5 * 1024], ids=["64chunks", "ragged_partial_last", "one_chunk_rank1_empty", "5chunks"])
def test_two_ranks_gather_every_steps_h_out(nbytes):
    chain = __import__("importlib").import_module("hot-proofs-blake3-circom_amd.chain")
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, T.free_port(), nbytes, ret), nprocs=2, join=True)
    data = T.workloads().lcg_preimage(nbytes, seed=1).tobytes()
    n = (nbytes + 1023) // 1024
    want = np.array([r for c in range(n) for r in _leaf_h_out(data, c)], dtype=np.uint32)
    assert want.shape[0] == T.pkg().lib().b3w_chain_num_leaf_steps(nbytes)
    n_par = T.pkg().lib().b3w_chain_parent_row(n, n)
    for rank in (0, 1):
        leaf, par = ret[rank]
        assert np.array_equal(leaf, want), rank
        assert par.shape == (n_par, 8) and np.array_equal(par.reshape(-1), np.arange(n_par * 8, dtype=np.uint32)), rank
    # the last step of chunk c yields chunk c's chaining value (for a single chunk: the root hash)
    leaf = ret[1][0]
    row = 0
    for c in range(n):
        nb = len(_leaf_h_out(data, c))
        row += nb
        assert list(leaf[row - 1]) == B.chunk_cv(data[c * 1024:(c + 1) * 1024], c, n == 1), c
    sh = chain.step_shards(nbytes, 2)
    assert sum(x[2] for x in sh) == want.shape[0] and sum(x[4] for x in sh) == n_par


def test_step_shards_tile_the_pass():
    """contiguous chunk ranges: leaf rows and parent rows of consecutive ranks follow each other, whatever the chunk count"""
    chain = __import__("importlib").import_module("hot-proofs-blake3-circom_amd.chain")
    L = T.pkg().lib()
    for nbytes in (1, 64, 1024, 1025, 4096, 5 * 1024 + 1, 100 * 1024, (1 << 20), (1 << 20) + 77):
        n = L.b3w_chain_num_chunks(nbytes)
        for world in (1, 2, 3, 8):
            sh = chain.step_shards(nbytes, world)
            assert sh[0][0] == 0 and sh[0][3] == 0
            for a, b in zip(sh, sh[1:]):
                assert a[0] + a[1] == b[0] and a[3] + a[4] == b[3]
            assert sh[-1][0] + sh[-1][1] == n
            assert sum(x[2] for x in sh) == L.b3w_chain_num_leaf_steps(nbytes)
            assert sum(x[4] for x in sh) == L.b3w_chain_parent_row(n, n)
            f, k = np.zeros(1, np.uint64), np.zeros(1, np.uint32)
            for r in range(world):                               # the native shard rule is the same rule
                L.b3w_chain_shard(n, r, world, f.ctypes.data_as(__import__("ctypes").POINTER(__import__("ctypes").c_uint64)),
                                  k.ctypes.data_as(__import__("ctypes").POINTER(__import__("ctypes").c_uint32)))
                assert (int(f[0]), int(k[0])) == sh[r][:2]
