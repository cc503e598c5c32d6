"""Chained ("nova fold") mode: the device planner produces every step's input record from the preimage;
the nova kernels turn them into witnesses.  Checked the way the reference's Rust tests check the fold
(rust_fold/src/main.rs:414-539: "h_out after all steps == BLAKE3(input)") plus step-to-step chaining
and oracle parity of sampled bodies."""
import numpy as np
import pytest
import b3w_testlib as T
import blake3_ref as B

pytestmark = pytest.mark.gpu


def _run(m, data, circuit="nova_vesta"):
    import torch
    dev = torch.device("cuda:0")
    ctx = m.Context(circuit, 0)
    d_pre = torch.from_numpy(np.frombuffer(bytes(data), dtype=np.uint8).copy()).to(dev)
    plan = m.ChainPlanner(ctx).plan(d_pre)
    recs = plan["records"]
    n = recs.shape[0]
    d_bodies = torch.empty((n, ctx.body_bytes), dtype=torch.uint8, device=dev)
    d_pub = torch.zeros((n, 15), dtype=torch.int32, device=dev)
    d_st = torch.full((n,), -1, dtype=torch.int32, device=dev)
    ctx.run_device(recs.data_ptr(), n, d_bodies.data_ptr(), 0, d_pub.data_ptr(), d_st.data_ptr(),
                   torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return ctx, plan, recs.cpu().numpy().view(np.uint32), d_pub.cpu().numpy().view(np.uint32), d_st.cpu().numpy(), d_bodies


def _check_chain(data, plan, recs, pub, st):
    assert (st == 0).all()
    n_leaf, n_chunks, P = plan["n_leaf_steps"], plan["n_chunks"], plan["path_len"]
    cvs = plan["chunk_cvs"].cpu().numpy().view(np.uint32)
    root = plan["root"].cpu().numpy().view(np.uint32)
    assert list(root) == B.hash_words(data), "tree root != BLAKE3(preimage)"
    # leaf steps: step j of chunk c feeds step j+1 (z_{i+1} = public outputs of step i, blake3_circuit.rs:111-123)
    s = 0
    for c in range(n_chunks):
        nb = recs[s, 0]
        for j in range(nb):
            r, o = recs[s + j], pub[s + j]
            assert r[1] == j and r[10] == c and o[0] == nb and o[1] == j + 1
            if j + 1 < nb:
                assert np.array_equal(o[2:10], recs[s + j + 1][2:10]), (c, j)
                assert o[11] == r[14]                     # depth unchanged inside the chunk
            else:
                assert np.array_equal(o[2:10], cvs[c]), c   # chunk chaining value
                assert o[11] == (r[14] - 1 if r[14] > 0 else 0)
        s += nb
    assert s == n_leaf
    if n_chunks == 1:
        assert np.array_equal(pub[n_leaf - 1][2:10], root)
    if plan["n_parent_steps"]:
        rows = T.pkg().ChainPlanner(None).parent_rows(n_chunks)
        assert sum(pl for _, pl, _ in rows) == plan["n_parent_steps"]
        for c, (row, pl, provable) in enumerate(rows):
            base = n_leaf + row
            for j in range(pl):
                r, o = recs[base + j], pub[base + j]
                assert r[10] == c and r[12] == pl + 1 and r[13] == pl + 1
                assert r[14] == pl - 1 - j and r[31] == 64 and (r[23:31] == 0).all()
                if j + 1 < pl:
                    assert np.array_equal(o[2:10], recs[base + j + 1][2:10]), (c, j)
                else:
                    # the fold's final h_out == BLAKE3(input) exactly for the paths the circuit's index-bit rule gets right
                    assert np.array_equal(o[2:10], root) == provable, (c, provable)
                    assert o[11] == 0


@pytest.mark.parametrize("data", [bytes(4), bytes([117]) * 17, bytes(68), bytes(1024), bytes(1028), bytes(1024 * 3 + 5),
                                  bytes(range(256)) * 8, bytes(range(251)) * 33],
                         ids=["zero4", "b117x17", "zero68", "zero1024", "zero1028", "zero3077", "2048", "8283"])
def test_reference_rust_test_shapes(data):
    """The inputs of rust_fold/src/main.rs:478-539 (+ two more): every chunk's path at once."""
    m = T.pkg()
    ctx, plan, recs, pub, st, d_bodies = _run(m, data)
    _check_chain(data, plan, recs, pub, st)
    # sampled bodies against the oracle
    idx = sorted(set([0, recs.shape[0] // 2, recs.shape[0] - 1]))
    _, want = T.oracle_batch_u32("nova_vesta", recs[idx])
    for k, i in enumerate(idx):
        assert np.array_equal(d_bodies[i].cpu().numpy(), want[k]), i
    ctx.close()


@pytest.mark.parametrize("nbytes", [3 * 1024, 5 * 1024 + 1, 6 * 1024, 7 * 1024 - 3, 11 * 1024, 100 * 1024 + 77, 1, 37 * 1024 + 64])
def test_paths_for_any_chunk_count(nbytes):
    """Incomplete trees: every chunk gets the parent steps of ITS path (lengths differ per chunk), planned by the reference
    driver's rule; the tree root is BLAKE3(input); the paths whose position agrees with their index bits end in it, the
    others do not (as with the reference: tests/golden/incomplete_trees.nova_vesta.json)."""
    m = T.pkg()
    rng = np.random.default_rng(nbytes)
    data = rng.integers(0, 256, nbytes, dtype=np.uint8).tobytes()
    ctx, plan, recs, pub, st, _ = _run(m, data, "nova_bn254")
    n = plan["n_chunks"]
    rows = m.ChainPlanner(ctx).parent_rows(n)
    assert plan["n_parent_steps"] == sum(pl for _, pl, _ in rows) and [pl for _, pl, _ in rows] == [_path_len(c, n) for c in range(n)]
    assert all(ok for _, _, ok in rows[:1 << (n.bit_length() - 1)])          # the leading power-of-two subtree is always provable
    if n & (n - 1):
        assert not rows[-1][2]                                                # the last chunk of an incomplete tree never is
    _check_chain(data, plan, recs, pub, st)
    ctx.close()


def _path_len(c, n):
    p = 0
    while n > 1:
        k = 1
        while k * 2 < n:
            k *= 2
        if c < k:
            n = k
        else:
            c, n = c - k, n - k
        p += 1
    return p


def test_incomplete_trees_match_the_reference_wasm_transcript():
    """The reference WASM folded along every path of 2 ... 100-chunk trees by the reference driver's rules
    (tools/probe_incomplete_trees.js, build container): the planner's parent records equal the transcript's inputs word for
    word, the kernels' public outputs equal the WASM's, and `provable` is exactly "ended in BLAKE3(input)"."""
    import gzip, json, os
    m = T.pkg()
    W = T.workloads()
    doc = json.load(gzip.open(os.path.join(T.GOLD, "incomplete_trees.nova_vesta.json.gz"), "rt"))
    for tree in doc["trees"]:
        n = tree["n_chunks"]
        data = W.lcg_preimage(n * 1024, seed=1).tobytes()
        ctx, plan, recs, pub, st, _ = _run(m, data, "nova_vesta")
        assert (st == 0).all() and list(plan["root"].cpu().numpy().view(np.uint32)) == tree["root"]
        rows = m.ChainPlanner(ctx).parent_rows(n)
        n_leaf = plan["n_leaf_steps"]
        for leaf in tree["leaves"]:
            c = leaf["leaf"]
            row, pl, provable = rows[c]
            assert pl == leaf["path_len"] and provable == leaf["ends_in_root"] == leaf["bits_agree"], (n, c)
            steps = leaf["steps"]                         # the last leaf block, then the parent steps
            assert len(steps) == 1 + pl
            idx = [c * 16 + 15] + [n_leaf + row + j for j in range(pl)]
            for k, (i, stp) in enumerate(zip(idx, steps)):
                assert list(recs[i]) == stp["record"], (n, c, k)
                assert list(pub[i]) == stp["public"], (n, c, k)
            assert list(pub[idx[-1]][2:10]) == leaf["final_h"]
        ctx.close()


def test_config4_one_mib_preimage():
    """BASELINE config 4: 1 MiB preimage = LE stream of LCG(1): 1 024 chunks x 16 blocks = 16 384 leaf steps
    + 10 parent steps per chunk path, leaf_depth = total_depth = 11."""
    m = T.pkg()
    W = T.workloads()
    lcg = W.LCG(1)
    words = np.array([lcg.next() for _ in range(1 << 18)], dtype=np.uint32)
    data = words.tobytes()
    ctx, plan, recs, pub, st, d_bodies = _run(m, data)
    assert plan["n_leaf_steps"] == 16384 and plan["n_parent_steps"] == 10240 and plan["path_len"] == 10
    assert (recs[:16384, 12] == 11).all() and (recs[:16384, 13] == 11).all() and (recs[:16384, 14] == 10).all()
    _check_chain(data, plan, recs, pub, st)
    idx = np.array([0, 15, 16, 8191, 16383, 16384, 16393, 20000, 26623])
    _, want = T.oracle_batch_u32("nova_vesta", recs[idx])
    for k, i in enumerate(idx):
        assert np.array_equal(d_bodies[int(i)].cpu().numpy(), want[k]), i
    # every one of the 26 624 step witnesses against the step circuit's rank-1 constraints, on the device (independent of the
    # witness kernels and of the oracle: what synthesize_with_vec enforces for each step, rust_fold/src/utils.rs:17-88)
    import torch
    r1cs = m.R1cs(ctx)
    n = d_bodies.shape[0]
    d_viol = torch.full((n,), -1, dtype=torch.int32, device=d_bodies.device)
    r1cs.check_device(d_bodies.data_ptr(), n, 0, d_viol.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert n == 26624 and int(d_viol.abs().sum().item()) == 0
    r1cs.close()
    ctx.close()


def test_streamed_fold_8mib_ring_buffer():
    """Streaming driver (chain.fold_witnesses): 8 MiB preimage = 131 072 leaf + 106 496 parent steps = 177 GB of
    witness through a 2-deep ring of 16 384-step buffers, H2D of the preimage overlapped slice by slice.
    Final h_out of every chunk path == BLAKE3(preimage); a consumer sees every batch."""
    import torch
    m = T.pkg()
    rng = np.random.default_rng(2024)
    data = rng.integers(0, 256, 8 << 20, dtype=np.uint8)
    ctx = m.Context("nova_vesta", 0)
    seen = []

    def consumer(bodies, first_step, k):
        # slot 0 of every body is the constant 1: a device-side reduction the consumer enqueues per batch
        seen.append((first_step, k, (bodies[:, 0] == 1).sum()))
    out = m.chain.fold_witnesses(ctx, torch.from_numpy(data).pin_memory(), batch_steps=16384, ring=2, slice_chunks=1024,
                                 consumer=consumer)
    torch.cuda.synchronize()
    assert out["n_leaf_steps"] == 131072 and out["n_parent_steps"] == 8192 * 13 and out["path_len"] == 13
    assert (out["status"] == 0).all().item()
    assert sum(k for _, k, _ in seen) == 131072 + 8192 * 13 and all(int(c.item()) == k for _, k, c in seen)
    want = B.hash_words(data.tobytes())
    assert list(out["root"].cpu().numpy().view(np.uint32)) == want
    pub = out["public"].cpu().numpy().view(np.uint32)
    last_parent = pub[131072 + 12::13][:8192]          # step j = 12 (depth 0) of every chunk path
    assert (last_parent[:, 2:10] == np.array(want, dtype=np.uint32)).all()
    assert (last_parent[:, 11] == 0).all()
    ctx.close()


def test_config5_one_gib_preimage_full_size():
    """BASELINE config 5 at its full size: the 1 GiB little-endian stream of LCG(1) -> 1 048 576 chunks -> 16 777 216 leaf +
    20 971 520 parent steps = 28 TB of step witnesses through the two-deep ring, H2D overlapped slice by slice.
    Checked against an independent numpy BLAKE3 over the whole preimage: every chunk's leaf chain ends in its chaining
    value, every parent step carries the right sibling, every one of the 1 048 576 paths ends in BLAKE3(preimage), all
    statuses are 0; and 96 bodies copied out of the ring while it was being overwritten (both slots, first to last
    generation) equal the oracle's witnesses of their step records byte for byte."""
    import torch
    m = T.pkg()
    W = T.workloads()
    dev = torch.device("cuda:0")
    data = W.lcg_preimage(1 << 30, seed=1)
    n_chunks, P, batch = 1 << 20, 20, 16384
    n_leaf, n_par = 16 * n_chunks, P * n_chunks
    n_batches = n_leaf // batch + n_par // batch
    picks = set(np.linspace(0, n_batches - 1, 96).astype(np.int64).tolist())
    assert {b & 1 for b in picks} == {0, 1}                    # both ring slots
    ctx = m.Context("nova_vesta", 0)
    grabbed, seen = [], [0]

    def consumer(view, first_step, k):
        b = seen[0]
        seen[0] += 1
        if b in picks:
            j = (b * 7919) % k
            grabbed.append((first_step + j, view[j].clone()))      # enqueued on the pass's stream, before the slot is reused
    out = m.chain.fold_witnesses(ctx, torch.from_numpy(data).pin_memory(), batch_steps=batch, ring=2, consumer=consumer)
    torch.cuda.synchronize()
    assert (out["n_leaf_steps"], out["n_parent_steps"], out["path_len"], out["n_chunks"]) == (n_leaf, n_par, P, n_chunks)
    assert seen[0] == n_batches and len(grabbed) == 96
    assert int(out["status"].abs().sum().item()) == 0

    cvs = B.chunk_cvs_np(data)
    levels = B.tree_levels_np(cvs)
    root = levels[-1][0]
    assert list(out["root"].cpu().numpy().view(np.uint32)) == list(root)
    as_dev = lambda a: torch.from_numpy(a.view(np.int32)).to(dev)
    pub, recs = out["public"], out["records"]
    # leaf chains: block 15 of every chunk outputs the chunk's chaining value; its depth drops from 20 to 19
    last_leaf = pub[15:n_leaf:16]
    assert torch.equal(last_leaf[:, 2:10], as_dev(cvs)) and bool((last_leaf[:, 11] == P - 1).all().item())
    assert bool((recs[:n_leaf, 10].view(n_chunks, 16) == torch.arange(n_chunks, device=dev, dtype=torch.int32)[:, None]).all().item())
    # parent steps: step j of chunk c starts from the subtree value at height j and carries the sibling at that height
    par_recs = recs[n_leaf:].view(n_chunks, P, 32)
    c_idx = torch.arange(n_chunks, device=dev, dtype=torch.int64)
    for j in range(P):
        lv = as_dev(levels[j])
        node = c_idx >> j
        assert torch.equal(par_recs[:, j, 2:10], lv[node]), j
        assert torch.equal(par_recs[:, j, 15:23], lv[node ^ 1]), j
        assert bool((par_recs[:, j, 14] == P - 1 - j).all().item())
    # every path ends in BLAKE3(preimage)
    final = pub[n_leaf + P - 1::P]
    assert final.shape[0] == n_chunks and bool((final[:, 2:10] == as_dev(root.reshape(1, 8))).all().item())
    assert bool((final[:, 11] == 0).all().item())
    # bodies that went through the ring
    steps = [s for s, _ in grabbed]
    want_recs = recs[torch.tensor(steps, device=dev)].cpu().numpy().view(np.uint32)
    bad, want = T.oracle_batch_u32("nova_vesta", want_recs)
    assert bad == 0
    for k, (s, body) in enumerate(grabbed):
        assert np.array_equal(body.cpu().numpy(), want[k]), s
    # ... and satisfy every constraint of the nova step circuit (derived system of the Vesta O2 build, on the device)
    r1cs = m.R1cs(ctx)
    stack = torch.stack([b for _, b in grabbed])
    viol = torch.full((len(grabbed),), -1, dtype=torch.int32, device=dev)
    r1cs.check_device(stack.data_ptr(), len(grabbed), stack.stride(0), viol.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert int(viol.abs().sum().item()) == 0
    r1cs.close()
    assert min(steps) < batch and max(steps) >= n_leaf + n_par - batch and any(s < n_leaf for s in steps) and any(s >= n_leaf for s in steps)
    ctx.close()


def _two_rank_worker(rank, world, port, nbytes, ret):
    import os, torch, torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)      # both ranks share the one GPU of the test box
    try:
        m = T.pkg()
        torch.cuda.set_device(0)
        rng = np.random.default_rng(99)
        data = rng.integers(0, 256, nbytes, dtype=np.uint8)
        ctx = m.Context("nova_bn254", 0)
        out = m.chain.fold_witnesses(ctx, data, batch_steps=4096)
        torch.cuda.synchronize()
        ret[rank] = dict(root=out["root"].cpu().numpy().view(np.uint32).tolist(), first_chunk=out["first_chunk"],
                         n_local=out["n_chunks_local"], ok=bool((out["status"] == 0).all().item()),
                         last=out["public"].cpu().numpy().view(np.uint32)[out["n_leaf_steps"] + out["path_len"] - 1::out["path_len"], 2:10].tolist())
    finally:
        dist.destroy_process_group()


def test_two_ranks_shard_chunks_and_exchange_cvs():
    """N > 1: each rank folds its contiguous chunk range; the chunk CVs are all-gathered so both build the same
    tree; every chunk path on either rank ends in BLAKE3(preimage)."""
    import os
    import torch.multiprocessing as mp
    nbytes = 64 * 1024
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_two_rank_worker, args=(2, T.free_port(), nbytes, ret), nprocs=2, join=True)
    data = np.random.default_rng(99).integers(0, 256, nbytes, dtype=np.uint8).tobytes()
    want = B.hash_words(data)
    assert ret[0]["root"] == want and ret[1]["root"] == want and ret[0]["ok"] and ret[1]["ok"]
    assert (ret[0]["first_chunk"], ret[0]["n_local"], ret[1]["first_chunk"], ret[1]["n_local"]) == (0, 32, 32, 32)
    for r in (0, 1):
        assert len(ret[r]["last"]) == 32 and all(x == want for x in ret[r]["last"])


def _hout_worker(rank, world, port, nbytes, ret):
    import os, torch, torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)      # both ranks share the one GPU of the test box
    try:
        m = T.pkg()
        torch.cuda.set_device(0)
        data = m.workloads.lcg_preimage(nbytes, seed=1)
        ctx = m.Context("nova_vesta", 0)
        out = m.chain.fold_witnesses(ctx, data, batch_steps=4096)
        torch.cuda.synchronize()
        ret[rank] = dict(leaf=out["h_out_all"].cpu().numpy().view(np.uint32).copy(),
                         par=out["h_out_parents_all"].cpu().numpy().view(np.uint32).copy(),
                         root=out["root"].cpu().numpy().view(np.uint32).tolist(), n_leaf_local=out["n_leaf_steps"],
                         ok=bool((out["status"] == 0).all().item()))
        ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("nbytes", [1 << 20], ids=["config4_1mib"])      # (ragged shards over torch.distributed: tests/test_chain_hout_gloo.py on the CPU; natively: test_gpu_native_exchange.py)
def test_two_ranks_gather_h_out_of_every_step(nbytes):
    """BASELINE config 4 as it is worded: the 1 MiB LCG(1) preimage -> 16 384 chained leaf steps sharded over the ranks, and the
    per-step h_out (16 384 x 8 u32) all-gathered inside the pass: on BOTH ranks the gathered h_out of step 16 c + 15 is chunk c's
    chaining value for every chunk, and every provable chunk path's last parent step carries BLAKE3(preimage)
    (z_{i+1} = outputs of step i: rust_fold/src/blake3_circuit.rs:111-123)."""
    import torch.multiprocessing as mp
    m = T.pkg()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_hout_worker, args=(2, T.free_port(), nbytes, ret), nprocs=2, join=True)
    data = m.workloads.lcg_preimage(nbytes, seed=1)
    n = (nbytes + 1023) // 1024
    L = m.lib()
    want_root = B.hash_words(data.tobytes())
    full = n - (1 if nbytes % 1024 else 0)
    cvs = B.chunk_cvs_np(data[:full * 1024]) if full >= 2 else None
    n_leaf, n_par = L.b3w_chain_num_leaf_steps(nbytes), L.b3w_chain_parent_row(n, n)
    assert ret[0]["n_leaf_local"] + ret[1]["n_leaf_local"] == n_leaf and ret[0]["n_leaf_local"] > 0 and ret[1]["n_leaf_local"] > 0
    for r in (0, 1):
        assert ret[r]["ok"] and ret[r]["root"] == want_root
        leaf, par = ret[r]["leaf"], ret[r]["par"]
        assert leaf.shape == (n_leaf, 8) and par.shape == (n_par, 8)
        assert np.array_equal(leaf[15:full * 16:16], cvs), "gathered h_out of step 16c+15 != chunk CV c"
        if full < n:                                                      # the partial last chunk: its last block's h_out
            assert list(leaf[-1]) == B.chunk_cv(data[full * 1024:].tobytes(), full, False)
        for c in range(n):
            row = L.b3w_chain_parent_row(c, n)
            plen = L.b3w_chain_path_len(c, n)
            assert (list(par[row + plen - 1]) == want_root) == bool(L.b3w_chain_path_provable(c, n)), c
    assert np.array_equal(ret[0]["leaf"], ret[1]["leaf"]) and np.array_equal(ret[0]["par"], ret[1]["par"])
    # ... and equal to what ONE rank computes for the whole preimage
    ctx = m.Context("nova_vesta", 0)
    one = m.chain.fold_witnesses(ctx, data, batch_steps=4096)
    import torch
    torch.cuda.synchronize()
    assert one["h_out_all"].shape == (n_leaf, 8)
    assert np.array_equal(one["h_out_all"].cpu().numpy().view(np.uint32), ret[0]["leaf"])
    assert np.array_equal(one["h_out_parents_all"].cpu().numpy().view(np.uint32), ret[1]["par"])
    ctx.close()


def test_native_h_out_exchange_over_rccl_single_rank():
    """b3w_chain_run_parents_sharded + b3w_chain_allgather_hout (the C-ABI's own RCCL calls) with the one-rank communicator a
    one-GPU box can form: both exchanges run through ncclAllGather, the result is the pass's own h_out in step order; twice,
    so that the second pass runs on the exchange buffers the first one allocated."""
    import torch
    m = T.pkg()
    ctx = m.Context("nova_vesta", 0)
    comm = m.Comm(ctx, m.Comm.unique_id(), 0, 1)
    data = m.workloads.lcg_preimage(37 * 1024 + 333, seed=1)
    for _ in range(2):
        out = m.chain.fold_witnesses(ctx, data, batch_steps=256, comm=comm)
        torch.cuda.synchronize()
        pub = out["public"].cpu().numpy().view(np.uint32)
        assert bool((out["status"] == 0).all().item())
        assert out["root"].cpu().numpy().view(np.uint32).tolist() == B.hash_words(data.tobytes())
        nl = out["n_leaf_steps"]
        assert out["h_out_all"].shape == (nl, 8) and out["h_out_parents_all"].shape == (out["n_parent_steps"], 8)
        assert np.array_equal(out["h_out_all"].cpu().numpy().view(np.uint32), pub[:nl, 2:10])
        assert np.array_equal(out["h_out_parents_all"].cpu().numpy().view(np.uint32), pub[nl:, 2:10])
    # a chain that is not this rank's shard is refused
    import ctypes
    L = m.lib()
    h = ctypes.c_void_p()
    assert L.b3w_chain_create(ctx.handle, 8192, 1, 2, 64, 2, 1, ctypes.byref(h)) == 0
    d = torch.zeros(8 * 128, dtype=torch.int32, device="cuda")
    assert L.b3w_chain_allgather_hout(h, comm.handle, d.data_ptr(), None, None) == 100 and "b3w_chain_shard" in ctx.last_error()
    assert L.b3w_chain_allgather_hout(h, comm.handle, None, None, None) == 100
    L.b3w_chain_destroy(h)
    comm.close()
    ctx.close()


def test_ring_spares_are_bounded_and_trimmed():
    """a context keeps the ring buffers of a destroyed chain for the next chain of the same geometry — one size at a
    time, and b3w_ctx_trim releases them."""
    import ctypes
    m = T.pkg()
    L = m.lib()
    ctx = m.Context("nova_vesta", 0)
    live = lambda: ctx.bodies_stats()["live_buffers"]
    base = live()
    h = ctypes.c_void_p()
    assert L.b3w_chain_create(ctx.handle, 1 << 16, 0, 64, 1024, 2, 1, ctypes.byref(h)) == 0        # 2 x 1 024 bodies = 2 x 763 MB (placed)
    assert live() == base + 2
    L.b3w_chain_destroy(h)
    assert live() == base + 2                                   # kept as spares
    assert L.b3w_chain_create(ctx.handle, 1 << 16, 0, 64, 1024, 2, 1, ctypes.byref(h)) == 0        # same geometry: reused
    assert live() == base + 2
    L.b3w_chain_destroy(h)
    assert L.b3w_chain_create(ctx.handle, 1 << 16, 0, 64, 768, 2, 1, ctypes.byref(h)) == 0         # another geometry: the old spares go first
    assert live() == base + 2
    L.b3w_chain_destroy(h)
    assert live() == base + 2
    assert L.b3w_ctx_trim(ctx.handle) == 0 and live() == base
    assert L.b3w_ctx_trim(None) == 100
    ctx.close()


def test_native_chain_driver_argument_errors():
    """b3w_chain_* refuses what it cannot do: compression contexts, chunk ranges past the preimage, a chunk sub-range
    without the other ranks' chaining values, zero-sized rings."""
    import ctypes
    m = T.pkg()
    L = m.lib()
    h = ctypes.c_void_p()
    comp = m.Context("compression", 0)
    assert L.b3w_chain_create(comp.handle, 4096, 0, 4, 64, 2, 1, ctypes.byref(h)) == 100 and "nova" in comp.last_error()
    comp.close()
    ctx = m.Context("nova_vesta", 0)
    assert L.b3w_chain_create(ctx.handle, 4096, 2, 3, 64, 2, 1, ctypes.byref(h)) == 100          # 4 chunks: [2, 5) is past the end
    assert L.b3w_chain_create(ctx.handle, 4096, 0, 4, 0, 2, 1, ctypes.byref(h)) == 100
    assert L.b3w_chain_create(ctx.handle, 4096, 0, 4, 64, 0, 1, ctypes.byref(h)) == 100
    assert L.b3w_chain_create(ctx.handle, 4096, 1, 2, 64, 2, 1, ctypes.byref(h)) == 0            # a rank's share: chunks 1..2
    data = (np.arange(4096) % 251).astype(np.uint8)
    assert L.b3w_chain_run_leaves(h, data.ctypes.data, None, None, None) == 0
    assert L.b3w_chain_run_parents(h, None, None, None, None) == 100 and "all ranks" in ctx.last_error()
    L.b3w_chain_destroy(h)
    p, pl = ctypes.c_void_p(), ctypes.c_int32()
    assert L.b3w_bodies_alloc(ctx.handle, 0, ctypes.byref(p), ctypes.byref(pl)) == 100
    assert L.b3w_bodies_free(ctx.handle, None) == 0
    ctx.close()
