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
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(32500 + os.getpid() % 2000), RANK="0", WORLD_SIZE="1")
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
