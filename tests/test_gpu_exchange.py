"""The N > 1 exchange on the real backend: RCCL ("nccl") with the one GPU a test box has (world_size 1) — exercises
the asynchronous all-gather + stream-level waits of sharding.PublicExchange around the real witness kernel."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist

import b3w_testlib as T

pytestmark = pytest.mark.gpu


def test_public_exchange_over_rccl_single_rank():
    m = T.pkg()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(T.free_port()), RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        n = 512
        ctx = m.Context("compression", 0)
        ex = m.sharding.PublicExchange(n, ctx.public_words, dev)
        assert ex.active and not ex.staged
        d_bodies = torch.empty(n * ctx.body_bytes, dtype=torch.uint8, device=dev)
        d_st = torch.zeros(n, dtype=torch.int32, device=dev)
        s = torch.cuda.current_stream().cuda_stream
        want = None
        for step in range(6):                                   # different records every step
            recs = m.workloads.config2_compression(n, first=1000 * step)
            d_recs = torch.from_numpy(recs.view(np.int32)).to(dev)
            pub = ex.next_buffer()
            ctx.run_device(d_recs.data_ptr(), n, d_bodies.data_ptr(), 0, pub.data_ptr(), d_st.data_ptr(), s)
            ex.post()
            if step == 5:
                _, bodies = T.oracle_batch_u32("compression", recs[:8])
                want = bodies.reshape(8, -1, 32)[:, 1:17, :4].copy().view(np.uint32).reshape(8, 16)
        allpub = ex.finish()
        torch.cuda.synchronize()
        assert allpub.shape == (n, 16)
        assert np.array_equal(allpub[:8].cpu().numpy().view(np.uint32), want)
        ctx.close()
    finally:
        dist.destroy_process_group()


def test_native_rccl_allgather_single_rank():
    """b3w_comm_* (the C-ABI's own RCCL exchange, for hosts without torch.distributed — Node, C, Rust): a one-rank
    communicator on the test box's GPU gathers a batch's public outputs; librccl is loaded at run time."""
    import ctypes
    m = T.pkg()
    L = m.lib()
    ctx = m.Context("compression", 0)
    uid = (ctypes.c_uint8 * 128)()
    assert L.b3w_comm_unique_id(uid) == 0, ctx.last_error()
    comm = ctypes.c_void_p()
    assert L.b3w_comm_create(ctx.handle, uid, 0, 1, ctypes.byref(comm)) == 0, ctx.last_error()
    n = 256
    recs = m.workloads.config2_compression(n, first=31)
    dev = torch.device("cuda", 0)
    d_recs = torch.from_numpy(recs.view(np.int32)).to(dev)
    d_pub = torch.zeros((n, 16), dtype=torch.int32, device=dev)
    d_all = torch.full((n, 16), -1, dtype=torch.int32, device=dev)
    d_bodies = torch.empty(n * ctx.body_bytes, dtype=torch.uint8, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    ctx.run_device(d_recs.data_ptr(), n, d_bodies.data_ptr(), 0, d_pub.data_ptr(), 0, s)
    assert L.b3w_comm_allgather(comm, d_pub.data_ptr(), d_all.data_ptr(), n * 16 * 4, s) == 0, ctx.last_error()
    torch.cuda.synchronize()
    assert torch.equal(d_all, d_pub)
    _, bodies = T.oracle_batch_u32("compression", recs[:4])
    want = bodies.reshape(4, -1, 32)[:, 1:17, :4].copy().view(np.uint32).reshape(4, 16)
    assert np.array_equal(d_all[:4].cpu().numpy().view(np.uint32), want)
    assert L.b3w_comm_create(ctx.handle, uid, 3, 2, ctypes.byref(ctypes.c_void_p())) == 100      # rank outside [0, nranks)
    L.b3w_comm_destroy(comm)
    ctx.close()
