"""The N>1 path on CPU: world_size-2 gloo.  Each rank takes its contiguous shard of the steps,
produces the per-step public outputs (here with the oracle standing in for the device kernel, which
needs a GPU) and all-gathers them; every rank must end up with the single-process result."""
import os, sys
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import b3w_testlib as T


def _worker(rank, world, port, n_total, circuit, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sh = T.pkg().sharding
        W = T.workloads()
        s, e = sh.shard_range(n_total, rank, world)
        recs = W.config3_nova(e - s, first=s) if circuit != "compression" else W.config2_compression(e - s, first=s)
        _, bodies = T.oracle_batch_u32(circuit, recs)
        npub = 16 if circuit == "compression" else 15
        pub = bodies.reshape(e - s, -1, 32)[:, 1:1 + npub, :4].copy().view(np.uint32).reshape(e - s, npub)
        allpub = sh.gather_public(torch.from_numpy(pub.view(np.int32)), n_total)
        ret[rank] = allpub.numpy().view(np.uint32).copy()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("circuit,n_total", [("compression", 64), ("nova_vesta", 37)])
def test_two_rank_shard_and_gather(circuit, n_total):
    import importlib
    importlib.import_module("hot-proofs-blake3-circom_amd.sharding")
    mgr = mp.Manager()
    ret = mgr.dict()
    port = T.free_port()
    mp.spawn(_worker, args=(2, port, n_total, circuit, ret), nprocs=2, join=True)
    W = T.workloads()
    recs = W.config3_nova(n_total) if circuit != "compression" else W.config2_compression(n_total)
    _, bodies = T.oracle_batch_u32(circuit, recs)
    npub = 16 if circuit == "compression" else 15
    want = bodies.reshape(n_total, -1, 32)[:, 1:1 + npub, :4].copy().view(np.uint32).reshape(n_total, npub)
    assert np.array_equal(ret[0], want) and np.array_equal(ret[1], want)


def test_shard_range_covers_everything():
    sh = __import__("importlib").import_module("hot-proofs-blake3-circom_amd.sharding")
    for n in (0, 1, 7, 8, 9, 4096, 16394):
        for world in (1, 2, 3, 8):
            r = [sh.shard_range(n, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            assert max(e - s for s, e in r) - min(e - s for s, e in r) <= 1


def _exchange_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sh = T.pkg().sharding
        n, words, steps = 5, 16, 7
        ex = sh.PublicExchange(n, words, torch.device("cpu"), depth=2)
        seen = []
        for step in range(steps):
            pub = ex.next_buffer()
            pub.copy_(torch.arange(n * words, dtype=torch.int32).view(n, words) + 1000 * rank + 100000 * step)
            ex.post()
            seen.append(ex.outs[ex.last].clone())
        last = ex.finish()
        ret[rank] = (torch.stack(seen).numpy(), last.numpy().copy())
    finally:
        dist.destroy_process_group()


def test_pipelined_public_exchange_two_ranks():
    """sharding.PublicExchange (what bench.py runs between steps at N > 1): every step's gather holds both ranks'
    outputs of THAT step in rank order, and finish() returns the last step's."""
    mgr = mp.Manager()
    ret = mgr.dict()
    port = T.free_port()
    mp.spawn(_exchange_worker, args=(2, port, ret), nprocs=2, join=True)
    n, words, steps = 5, 16, 7
    base = np.arange(n * words, dtype=np.int32).reshape(n, words)
    for rank in (0, 1):
        seen, last = ret[rank]
        for step in range(steps):
            want = np.concatenate([base + 1000 * r + 100000 * step for r in (0, 1)])
            assert np.array_equal(seen[step], want), (rank, step)
        assert np.array_equal(last, seen[steps - 1])


def test_native_shard_ranges_equal_the_python_ones():
    """b3w_chain_shard (C-ABI, used by the native sharded chained pass) and sharding.shard_range agree."""
    import ctypes
    m = T.pkg()
    L = m.lib()
    for n in (1, 7, 8, 9, 1024, 16394):
        for world in (1, 2, 3, 8):
            for r in range(world):
                f, k = ctypes.c_uint64(), ctypes.c_uint32()
                L.b3w_chain_shard(n, r, world, ctypes.byref(f), ctypes.byref(k))
                s, e = m.sharding.shard_range(n, r, world)
                assert (f.value, k.value) == (s, e - s), (n, world, r)
