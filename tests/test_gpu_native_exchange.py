"""The NATIVE exchange of the sharded chained pass with more than one rank (round-3 verdict, weak #3): b3w_chain_run_parents_sharded,
b3w_chain_allgather_hout[_host] and b3w_batch_allgather_public through a b3w_comm whose transport needs no second GPU —
b3w_comm_create_host (POSIX shared memory between the ranks' processes) and b3w_comm_create_external (the caller's collective:
torch.distributed over gloo, staged through the host).  Two and three ranks share the test box's one GPU; what every rank
gathers must equal what ONE rank computes for the whole preimage:
  1 MiB         BASELINE config 4: 1 024 chunks, 16 384 leaf steps, complete tree (at three ranks: shards of 342 / 341 / 341 chunks)
  5 KiB + 100   six chunks, a partial last chunk, an incomplete tree: ragged leaf AND parent shards, padding on the wire
  700 B         one chunk: ranks 1.. have nothing (the empty-shard rows of the scatter table)
This is the reference's z_{i+1} = public outputs of step i (rust_fold/src/blake3_circuit.rs:111-123, fed back at
rust_fold/src/main.rs:166-179), for every step of every rank."""
import ctypes
import os
import uuid

import numpy as np
import pytest

import b3w_testlib as T
import blake3_ref as B

pytestmark = pytest.mark.gpu

SHAPES = [("config4_1mib", 1 << 20), ("6chunks_partial_last", 5 * 1024 + 100), ("one_chunk", 700)]
if os.environ.get("B3W_TEST_EXTRA_PREIMAGE_BYTES"):           # a one-off soak shape (tools/jobs/r04/job31_soak.sh: 8 MiB + 77: several slices a rank, ragged)
    SHAPES.append(("extra", int(os.environ["B3W_TEST_EXTRA_PREIMAGE_BYTES"])))


def _worker(rank, world, transport, name, port, ret):
    import torch
    os.environ["B3W_PLACEMENT"] = "plain"                    # three processes on one card: no placement searches side by side
    os.environ["B3W_HOSTCOMM_TIMEOUT_S"] = "90"
    m = T.pkg()
    L = m.lib()
    torch.cuda.set_device(0)
    ctx = m.Context("nova_vesta", 0)
    if transport == "host":
        comm = m.Comm.host(ctx, name, rank, world)
    else:
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        comm = m.Comm.external(ctx, rank, world, m.sharding.torch_allgather(torch.device("cuda", 0)))
    assert (L.b3w_comm_rank(comm.handle), L.b3w_comm_size(comm.handle)) == (rank, world)
    res = {}
    try:
        for tag, nbytes in SHAPES:
            data = m.workloads.lcg_preimage(nbytes, seed=1)
            for rep in range(2):                             # the second pass runs on the exchange buffers the first one allocated
                out = m.chain.fold_witnesses(ctx, data, batch_steps=1024, comm=comm)
                torch.cuda.synchronize()
            cv_ms, h_ms = m.chain.exchange_ms(out)
            # the same exchange into host arrays (what the Node binding calls)
            h = ctx._chain_cache[next(iter(ctx._chain_cache))]
            n_leaf, n = L.b3w_chain_num_leaf_steps(nbytes), L.b3w_chain_num_chunks(nbytes)
            n_par = L.b3w_chain_parent_row(n, n)
            hl = np.zeros((n_leaf, 8), dtype=np.uint32)
            hp = np.zeros((max(n_par, 1), 8), dtype=np.uint32)
            rc = L.b3w_chain_allgather_hout_host(h, comm.handle, hl.ctypes.data, hp.ctypes.data if n_par else None, None)
            assert rc == 0, ctx.last_error()
            res[tag] = dict(leaf=out["h_out_all"].cpu().numpy().view(np.uint32).copy(), par=out["h_out_parents_all"].cpu().numpy().view(np.uint32).copy(),
                            leaf_host=hl, par_host=hp[:n_par], root=out["root"].cpu().numpy().view(np.uint32).tolist(),
                            first_chunk=out["first_chunk"], n_local=out["n_chunks_local"], n_leaf_local=out["n_leaf_steps"],
                            n_par_local=out["n_parent_steps"], ok=bool((out["status"] == 0).all().item()), ms=(cv_ms, h_ms))
        # batch mode: every rank's public outputs of its own 96 compression witnesses, on every rank
        cctx = m.Context("compression", 0)
        ccomm = m.Comm.host(cctx, name + "_b", rank, world) if transport == "host" else \
            m.Comm.external(cctx, rank, world, m.sharding.torch_allgather(torch.device("cuda", 0)))
        b = m.Batch(cctx, 96)
        b.run(m.workloads.config2_compression(96, first=96 * rank))
        allpub = np.zeros((world * 96, 16), dtype=np.uint32)
        assert L.b3w_batch_allgather_public(b.handle, ccomm.handle, allpub.ctypes.data) == 0, cctx.last_error()
        res["batch_public"] = allpub
        # a failing collective surfaces as B3W_E_RCCL with the reason, and the library stays usable
        if transport == "external":
            def boom(*a):
                raise RuntimeError("collective down")
            bad = m.Comm.external(cctx, rank, world, boom)
            with pytest.raises(m.B3WError) as ei:
                bad.allgather(b.device_ptr()[0], b.device_ptr()[0], 16, 0)
            assert ei.value.status == 105 and "collective down" in str(ei.value)
            bad.close()
        ccomm.close(); b.close(); cctx.close()
        ret[rank] = res
    finally:
        comm.close()
        ctx.close()
        if transport != "host":
            import torch.distributed as dist
            dist.destroy_process_group()


_one_rank = {}


def one_rank(nbytes):
    """what a single rank computes for the whole preimage (no communicator)"""
    if nbytes not in _one_rank:
        import torch
        m = T.pkg()
        ctx = m.Context("nova_vesta", 0)
        data = m.workloads.lcg_preimage(nbytes, seed=1)
        out = m.chain.fold_witnesses(ctx, data, batch_steps=1024)
        torch.cuda.synchronize()
        assert out["root"].cpu().numpy().view(np.uint32).tolist() == B.hash_words(data.tobytes())
        _one_rank[nbytes] = (out["h_out_all"].cpu().numpy().view(np.uint32).copy(), out["h_out_parents_all"].cpu().numpy().view(np.uint32).copy(),
                             out["root"].cpu().numpy().view(np.uint32).tolist())
        ctx.close()
    return _one_rank[nbytes]


@pytest.mark.parametrize("world,transport", [(2, "host"), (3, "host"), (2, "external")], ids=["host_x2", "host_x3", "external_gloo_x2"])
def test_native_exchange_with_several_ranks_equals_one_rank(world, transport):
    import torch.multiprocessing as mp
    m = T.pkg()
    L = m.lib()
    mgr = mp.Manager()
    ret = mgr.dict()
    name = "/b3w_t_" + uuid.uuid4().hex[:16]
    mp.spawn(_worker, args=(world, transport, name, T.free_port(), ret), nprocs=world, join=True)
    for tag, nbytes in SHAPES:
        leaf1, par1, root1 = one_rank(nbytes)
        n = L.b3w_chain_num_chunks(nbytes)
        total_leaf = total_par = 0
        for r in range(world):
            got = ret[r][tag]
            f, k = ctypes.c_uint64(), ctypes.c_uint32()
            L.b3w_chain_shard(n, r, world, ctypes.byref(f), ctypes.byref(k))
            assert (got["first_chunk"], got["n_local"]) == (f.value, k.value), (tag, r)
            assert got["ok"] and got["root"] == root1, (tag, r)
            assert got["leaf"].shape == leaf1.shape and got["par"].shape == par1.shape, (tag, r)
            assert np.array_equal(got["leaf"], leaf1), f"{tag}: rank {r} of {world} gathered other leaf h_out than one rank computes"
            assert np.array_equal(got["par"], par1), f"{tag}: rank {r} of {world} gathered other parent h_out than one rank computes"
            assert np.array_equal(got["leaf_host"], leaf1) and np.array_equal(got["par_host"], par1), (tag, r)
            total_leaf += got["n_leaf_local"]; total_par += got["n_par_local"]
            assert got["ms"][0] > 0 and got["ms"][1] > 0
        assert (total_leaf, total_par) == (leaf1.shape[0], par1.shape[0]), tag
    # the geometries the verdict asked for did occur: ragged shards, a rank without a chunk
    assert [ret[r]["6chunks_partial_last"]["n_local"] for r in range(world)] == ([3, 3] if world == 2 else [2, 2, 2])
    assert [ret[r]["one_chunk"]["n_local"] for r in range(world)] == [1] + [0] * (world - 1)
    if world == 3:
        assert [ret[r]["config4_1mib"]["n_local"] for r in range(3)] == [342, 341, 341]
    # batch mode: rank r's rows are the public outputs of ITS records
    want = np.concatenate([T.oracle_batch_u32("compression", m.workloads.config2_compression(96, first=96 * r))[1]
                           .reshape(96, -1, 32)[:, 1:17, :4].copy().view(np.uint32).reshape(96, 16) for r in range(world)])
    for r in range(world):
        assert np.array_equal(ret[r]["batch_public"], want), r


def test_unpack_scatter_of_ragged_blocks_without_a_second_process():
    """the scatter kernel behind b3w_chain_allgather_hout fed a hand-made receive buffer for three ranks — per-shard
    passes of ONE process packed the way the wire carries them (leaf rows, then parent rows, both padded to the largest shard) —
    through a communicator whose all-gather is a Python function that ignores the send buffer and writes that image."""
    import torch
    m = T.pkg()
    L = m.lib()
    nbytes, world = 5 * 1024 + 100, 3
    leaf1, par1, _ = one_rank(nbytes)
    sh = m.chain.step_shards(nbytes, world)
    mx_leaf, mx_par = max(max(x[2] for x in sh), 1), max(x[4] for x in sh)
    block = (mx_leaf + mx_par) * 8
    image = np.full((world, block), 0xDEADBEEF, dtype=np.uint32)      # padding that must not surface anywhere
    for r, (f, k, nleaf, p0, npar) in enumerate(sh):
        image[r, :nleaf * 8] = leaf1[f * 16:f * 16 + nleaf].reshape(-1)
        image[r, mx_leaf * 8:mx_leaf * 8 + npar * 8] = par1[p0:p0 + npar].reshape(-1)
    ctx = m.Context("nova_vesta", 0)
    dev = torch.device("cuda", 0)
    d_image = torch.from_numpy(image.view(np.int32)).to(dev)
    cv_all = torch.from_numpy(np.ascontiguousarray(leaf1[[min(16 * c + 15, leaf1.shape[0] - 1) for c in range(6)]]).view(np.int32)).to(dev)

    class _Dev:
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i4", "data": (ptr, False), "version": 2, "strides": None}

    calls = []

    def fake(d_send, d_recv, nbytes_per_rank, stream):
        torch.cuda.synchronize()
        calls.append(nbytes_per_rank)
        recv = torch.as_tensor(_Dev(d_recv, nbytes_per_rank * world // 4), device=dev)
        if nbytes_per_rank == block * 4:                     # the h_out exchange: the hand-made image
            recv.copy_(d_image.reshape(-1))
        else:                                                # the chunk-CV exchange: 2 chunks per rank, no padding at 6 chunks / 3 ranks
            recv.copy_(cv_all.reshape(-1))
        torch.cuda.synchronize()
    for rank in range(world):
        comm = m.Comm.external(ctx, rank, world, fake)
        out = m.chain.fold_witnesses(ctx, m.workloads.lcg_preimage(nbytes, seed=1), batch_steps=256, comm=comm)
        torch.cuda.synchronize()
        assert np.array_equal(out["h_out_all"].cpu().numpy().view(np.uint32), leaf1), rank
        assert np.array_equal(out["h_out_parents_all"].cpu().numpy().view(np.uint32), par1), rank
        comm.close()
    assert calls == [2 * 32, block * 4] * world
    ctx.close()


@pytest.mark.parametrize("transport", ["host", "external"])
def test_eight_ranks_config4_equal_shards_in_one_process(transport):
    """The target geometry of BASELINE config 4 — 8 ranks, 1 024 chunks in equal shards of 128, the "gathered where they lie, ONE
    collective" path of b3w_chain_run_parents_sharded (the chunk CVs land in level 0 of every rank's tree) and the h_out exchange of
    8 x 2 048 leaf + 8 x 1 280 parent rows.  A GPU box of this pool admits six processes to its card, so the eight ranks are eight
    THREADS of this process, each with its own context, chain, stream and communicator: the host shared-memory transport (every
    thread maps the segment, as processes would) and an external collective (a barrier-and-copy all-gather between the threads).
    Every rank's gathered arrays equal what one rank computes for the whole preimage."""
    import threading
    import torch
    m = T.pkg()
    L = m.lib()
    world, nbytes = 8, 1 << 20
    leaf1, par1, root1 = one_rank(nbytes)
    data = m.workloads.lcg_preimage(nbytes, seed=1)
    dev = torch.device("cuda", 0)
    name = "/b3w_t8_" + uuid.uuid4().hex[:16]
    os.environ["B3W_PLACEMENT"] = "plain"                    # eight ring pairs of 1.5 GB side by side: no placement searches
    os.environ["B3W_HOSTCOMM_TIMEOUT_S"] = "90"
    bar = threading.Barrier(world, timeout=120)
    stage = {}                                               # the external collective's meeting place: rank -> its send block

    class _Dev:
        def __init__(self, ptr, nb):
            self.__cuda_array_interface__ = {"shape": (nb,), "typestr": "|u1", "data": (ptr, False), "version": 2, "strides": None}

    def make_allgather(rank):
        def fn(d_send, d_recv, nb, stream):
            torch.cuda.synchronize()                         # (a test collective: ordered by waiting, not by the stream)
            stage[rank] = torch.as_tensor(_Dev(d_send, nb), device=dev).clone()
            bar.wait()
            recv = torch.as_tensor(_Dev(d_recv, nb * world), device=dev)
            for r in range(world):
                recv[r * nb:(r + 1) * nb].copy_(stage[r])
            torch.cuda.synchronize()
            bar.wait()
        return fn
    res, errs = {}, []

    def rank_main(rank):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                ctx = m.Context("nova_vesta", 0)
                comm = m.Comm.host(ctx, name, rank, world) if transport == "host" else m.Comm.external(ctx, rank, world, make_allgather(rank))
                try:
                    for rep in range(2):                     # the second pass runs on the exchange buffers the first one allocated
                        out = m.chain.fold_witnesses(ctx, data, batch_steps=1024, comm=comm)
                        torch.cuda.current_stream().synchronize()
                    res[rank] = dict(leaf=out["h_out_all"].cpu().numpy().view(np.uint32).copy(), par=out["h_out_parents_all"].cpu().numpy().view(np.uint32).copy(),
                                     root=out["root"].cpu().numpy().view(np.uint32).tolist(), first=out["first_chunk"], n_local=out["n_chunks_local"],
                                     n_leaf=out["n_leaf_steps"], n_par=out["n_parent_steps"], ok=bool((out["status"] == 0).all().item()))
                finally:
                    comm.close()
                    ctx.close()
        except BaseException as e:                           # a rank that dies must not leave the others at the barrier for good
            errs.append((rank, repr(e)))
            bar.abort()
    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    os.environ.pop("B3W_PLACEMENT", None)
    assert not errs, errs
    for r in range(world):
        got = res[r]
        assert (got["first"], got["n_local"], got["n_leaf"], got["n_par"]) == (128 * r, 128, 2048, 1280), r
        assert got["ok"] and got["root"] == root1, r
        assert np.array_equal(got["leaf"], leaf1), f"rank {r} of 8 gathered other leaf h_out than one rank computes"
        assert np.array_equal(got["par"], par1), f"rank {r} of 8 gathered other parent h_out than one rank computes"
