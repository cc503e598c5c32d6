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
    with torch.cuda.graph(g, stream=side):
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
    with torch.cuda.graph(g, stream=side):
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
