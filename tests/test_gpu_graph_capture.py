"""The device-pointer entry points only enqueue work on the caller's stream (no allocation, no synchronisation once their
buffers exist), so a caller can capture a launch-bound loop of small batches into a hipGraph and replay it: 62 us kernels
(256 witnesses) stop paying a host launch each.  Captured here through torch.cuda.CUDAGraph: four witness launches over
different records, the tamper check and the rank-1 constraint check; replayed on new inputs; results equal direct calls."""
import numpy as np
import pytest

import b3w_testlib as T

pytestmark = pytest.mark.gpu


def test_witness_and_checks_replay_from_a_graph():
    import torch
    m = T.pkg()
    dev = torch.device("cuda:0")
    ctx = m.Context("compression", 0)
    r1cs = m.R1cs(ctx)
    n, k = 256, 4
    recs = torch.zeros((k, n, 28), dtype=torch.int32, device=dev)
    bodies = torch.zeros((k, n, ctx.body_bytes), dtype=torch.uint8, device=dev)
    pub = torch.zeros((k, n, 16), dtype=torch.int32, device=dev)
    st = torch.full((k, n), -1, dtype=torch.int32, device=dev)
    mm = torch.full((k, n), -1, dtype=torch.int32, device=dev)
    viol = torch.full((k, n), -1, dtype=torch.int32, device=dev)

    def enqueue(stream):
        for i in range(k):
            ctx.run_device(recs[i].data_ptr(), n, bodies[i].data_ptr(), 0, pub[i].data_ptr(), st[i].data_ptr(), stream)
            ctx.verify_device(bodies[i].data_ptr(), n, 0, mm[i].data_ptr(), stream)
            r1cs.check_device(bodies[i].data_ptr(), n, 0, viol[i].data_ptr(), 0, stream)

    W = T.workloads()
    first = W.config2_compression(k * n, first=1000).reshape(k, n, 28)
    recs.copy_(torch.from_numpy(first.view(np.int32)))
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        enqueue(side.cuda_stream)                              # warm-up outside the capture (module load, first-use set-up)
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with T.pkg().graph_capture(g, stream=side):
        enqueue(torch.cuda.current_stream().cuda_stream)
    for round_, start in enumerate((5000, 9000)):              # replay on new inputs: only the record buffer changes
        new = W.config2_compression(k * n, first=start).reshape(k, n, 28)
        recs.copy_(torch.from_numpy(new.view(np.int32)))
        st.fill_(-1); mm.fill_(-1); viol.fill_(-1)
        g.replay()
        torch.cuda.synchronize()
        sums = [[int(x[i].abs().sum().item()) for i in range(k)] for x in (st, mm, viol)]
        assert sums == [[0] * k] * 3, sums
        idx = [(0, 0), (1, 17), (3, n - 1)]
        bad, want = T.oracle_batch_u32("compression", np.stack([new[i, j] for i, j in idx]))
        assert bad == 0
        for q, (i, j) in enumerate(idx):
            assert np.array_equal(bodies[i, j].cpu().numpy(), want[q]), (round_, i, j)
    r1cs.close(); ctx.close()


def test_first_check_on_a_capturing_stream_is_refused_not_allocated():
    """the first constraint check on a stream allocates that stream's deferred-row scratch; inside a capture that would
    break the capture — the call says so instead (and the capture goes on with what is allowed)."""
    import torch
    m = T.pkg()
    dev = torch.device("cuda:0")
    ctx = m.Context("compression", 0)
    r1cs = m.R1cs(ctx)
    n = 64
    recs = torch.from_numpy(T.workloads().config2_compression(n).view(np.int32)).to(dev)
    bodies = torch.zeros((n, ctx.body_bytes), dtype=torch.uint8, device=dev)
    viol = torch.full((n,), -1, dtype=torch.int32, device=dev)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        ctx.run_device(recs.data_ptr(), n, bodies.data_ptr(), 0, 0, 0, side.cuda_stream)          # (the witness kernel may be captured cold)
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with T.pkg().graph_capture(g, stream=side):
        s = torch.cuda.current_stream().cuda_stream
        ctx.run_device(recs.data_ptr(), n, bodies.data_ptr(), 0, 0, 0, s)
        with pytest.raises(m.B3WError) as e:
            r1cs.check_device(bodies.data_ptr(), n, 0, viol.data_ptr(), 0, s)
        assert e.value.status == 100 and "before capturing" in str(e.value)
    g.replay()
    torch.cuda.synchronize()
    r1cs.check_device(bodies.data_ptr(), n, 0, viol.data_ptr(), 0, 0)
    torch.cuda.synchronize()
    assert int(viol.abs().sum().item()) == 0
    r1cs.close(); ctx.close()


def test_scratch_of_short_lived_streams_is_bounded():
    """a caller that cycles through streams: a constraint system keeps the deferred-row scratch of at most eight of them"""
    import torch
    m = T.pkg()
    dev = torch.device("cuda:0")
    ctx = m.Context("compression", 0)
    r1cs = m.R1cs(ctx)
    n = 64
    recs = torch.from_numpy(T.workloads().config2_compression(n).view(np.int32)).to(dev)
    bodies = torch.zeros((n, ctx.body_bytes), dtype=torch.uint8, device=dev)
    ctx.run_device(recs.data_ptr(), n, bodies.data_ptr(), 0, 0, 0, 0)
    torch.cuda.synchronize()
    viol = torch.full((n,), -1, dtype=torch.int32, device=dev)
    free = []
    for k in range(24):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            r1cs.check_device(bodies.data_ptr(), n, 0, viol.data_ptr(), 0, st.cuda_stream)
        st.synchronize()
        assert int(viol.abs().sum().item()) == 0
        del st
        free.append(torch.cuda.mem_get_info()[0])
    # after the eighth stream the footprint stops growing (24 unbounded scratches would be 650 MB)
    assert free[8] - free[-1] < (64 << 20), [f >> 20 for f in free]
    r1cs.close(); ctx.close()


@pytest.mark.parametrize("world", [1, 8])
def test_a_whole_chained_pass_replays_from_a_graph(world):
    """r04 verdict 4b: the chained pass — copy of the preimage, leaf plan, witness batches through the ring, tree + parent plan on the
    chain's side stream, parent witnesses, and (world = 8: rank 0's share through the native sharded path) both exchanges through a
    communicator whose collective only enqueues device work — captured once per geometry into a hipGraph and replayed on NEW preimage
    bytes (the pinned host buffer the capture read from is overwritten in place): public outputs, root and the gathered h_out of every
    replay equal the eager pass over the same bytes.  The library allocates nothing and waits for nothing inside a pass that has run once;
    the forks to its side stream are joined before the run calls return, so the capture closes."""
    import ctypes
    import torch
    import blake3_ref as B
    m = T.pkg()
    dev = torch.device("cuda:0")
    ctx = m.Context("nova_vesta", 0)
    nbytes = 64 * 1024                                         # 64 chunks: 1 024 leaf + 384 parent steps (world 8: 128 + 48 here)
    host = torch.zeros(nbytes, dtype=torch.uint8).pin_memory()
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]

    def standin(d_send, d_recv, nb, stream):                   # this rank's block into place 0 of the gathered buffer: device work on `stream` only
        assert hip.hipMemcpyAsync(d_recv, d_send, nb, 3, stream) == 0
    comm = m.Comm.external(ctx, 0, world, standin) if world > 1 else None

    def fold():
        return m.chain.fold_witnesses(ctx, host, batch_steps=512, ring=2, comm=comm)

    def snapshot(out):
        return (out["public"].clone(), out["root"].clone(), out["h_out_all"].clone(), out["status"].clone())
    side = torch.cuda.Stream()
    data = [((np.arange(nbytes, dtype=np.uint64) * (7919 + 2 * k) + k) % 251).astype(np.uint8) for k in range(3)]
    want = []
    with torch.cuda.stream(side):
        for k in range(3):                                     # eager passes: what every replay must reproduce (and the warm-up of the capture)
            host.copy_(torch.from_numpy(data[k]))
            out = fold()
            side.synchronize()
            want.append(snapshot(out))
            if world == 1:
                assert out["root"].cpu().numpy().view(np.uint32).tolist() == B.hash_words(data[k].tobytes())
    g = torch.cuda.CUDAGraph()
    with T.pkg().graph_capture(g, stream=side):
        out = fold()
    for k in (1, 2, 0):
        host.copy_(torch.from_numpy(data[k]))
        for t in (out["public"], out["status"]):
            t.fill_(-1)
        g.replay()
        torch.cuda.synchronize()
        got = snapshot(out)
        own = slice(0, out["n_leaf_steps"])                    # (world 8: the gathered array's other places hold rank 0's block or nothing: only the own rows are compared)
        assert torch.equal(got[0], want[k][0]) and torch.equal(got[1], want[k][1]) and int(got[3].abs().sum().item()) == 0, k
        assert torch.equal(got[2][own], want[k][2][own]), k
    if comm is not None:
        comm.close()
    ctx.close()


def test_releasing_other_objects_while_a_capture_is_open_leaves_the_capture_valid():
    """r05: a finaliser may run at any time — Python's cyclic collector released a Context inside somebody's capture and the suite
    aborted (profiles/r05/gpu_suite_abort_in_capture.log).  The release paths now put their thread into the relaxed capture mode and wait
    for the device without hipDeviceSynchronize (csrc/b3w_capture.h).  Here: contexts, a batch, placed and plain body buffers, a constraint
    system with a used scratch, a commit key, a chain with ring buffers, a communicator are all released INSIDE a bare torch.cuda.graph
    capture of another context's launch; the capture closes, replays on new records, and the bodies are those records' witnesses."""
    import gc
    import torch
    m = T.pkg()
    dev = torch.device("cuda:0")
    n = 64
    ctx = m.Context("compression", 0)
    recs_a = T.workloads().config2_compression(n, first=3)
    recs_b = T.workloads().config2_compression(n, first=7000)
    d_recs = torch.from_numpy(recs_a.view(np.int32)).to(dev)
    d_bodies = torch.zeros((n, ctx.body_bytes), dtype=torch.uint8, device=dev)
    d_pub = torch.zeros((n, 16), dtype=torch.int32, device=dev)
    d_st = torch.full((n,), -1, dtype=torch.int32, device=dev)
    # the objects that will be released inside the capture, all of them used before
    victim = m.Context("nova_vesta", 0)
    v_r1cs = m.R1cs(victim)
    v_n = 32
    v_recs = torch.from_numpy(T.workloads().config3_nova(v_n).view(np.int32)).to(dev)
    v_placed = victim.alloc_bodies(v_n * victim.body_bytes)
    victim.run_device(v_recs.data_ptr(), v_n, v_placed.ptr, 0, 0, 0, 0)
    v_viol = torch.full((v_n,), -1, dtype=torch.int32, device=dev)
    v_r1cs.check_device(v_placed.ptr, v_n, 0, v_viol.data_ptr(), 0, 0)
    import ec_ref as E
    gens = E.points_to_bytes(E.random_points("pallas", 40, seed=b"capture"))
    v_key = m.CommitKey(victim, "pallas", gens, first_slot=victim.witness_size - 40, window=12)
    v_out = m.chain.fold_witnesses(victim, torch.zeros(8 * 1024, dtype=torch.uint8).pin_memory(), batch_steps=64, ring=2)
    v_batch = m.Batch(victim, 16)
    v_comm = m.Comm.external(victim, 0, 2, lambda a, b, c, d: None)
    torch.cuda.synchronize()
    assert int(v_viol.abs().sum().item()) == 0
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ctx.run_device(d_recs.data_ptr(), n, d_bodies.data_ptr(), 0, d_pub.data_ptr(), d_st.data_ptr(), side.cuda_stream)
    torch.cuda.synchronize()
    gc.collect()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):                     # (bare: the global capture mode, no collector games)
        ctx.run_device(d_recs.data_ptr(), n, d_bodies.data_ptr(), 0, d_pub.data_ptr(), d_st.data_ptr(), torch.cuda.current_stream().cuda_stream)
        del v_out
        v_comm.close(); v_batch.close()
        v_key.close(); v_r1cs.close(); v_placed.free()
        victim.close()                                         # (its cached chain and ring spares go with it)
        m.lib().b3w_bodies_trim()
    d_recs.copy_(torch.from_numpy(recs_b.view(np.int32)))
    d_bodies.zero_(); d_st.fill_(-1)
    g.replay()
    torch.cuda.synchronize()
    assert int(d_st.abs().sum().item()) == 0
    bad, want = T.oracle_batch_u32("compression", recs_b)
    assert bad == 0 and np.array_equal(d_bodies.cpu().numpy(), want)
    ctx.close()
