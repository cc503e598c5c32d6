"""Multi-GPU sharding of independent witnesses.

Units (compression instances / nova steps) are independent, so each rank takes a contiguous
range and witness bodies never leave the GPU that produced them.  The one exchange step is the
all-gather of the per-step public outputs (h_out ... : 15 or 16 u32 per step) that feeds the fold
(rust_fold/src/main.rs:166-179 consumes z_{i+1} = public outputs of step i).  One process per
GPU; torch.distributed backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.
"""
import torch
import torch.distributed as dist


def shard_range(n_total, rank, world):
    """Contiguous, balanced ranges: the first n_total % world ranks take one extra unit."""
    q, r = divmod(n_total, world)
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def gather_rows(local, counts, group=None):
    """All-gather of ragged row blocks: rank r contributes counts[r] rows (`local`: [counts[rank], words] on this rank, device or
    CPU).  Returns [sum(counts), words] in rank order on every rank.  Shards are padded to the largest for the collective
    (ncclAllGather takes equal counts) and trimmed afterwards."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local
    assert len(counts) == world and local.shape[0] == counts[dist.get_rank(group)]
    if local.is_cuda and dist.get_backend(group) == "gloo":
        # CPU rehearsal of the exchange (tests / single-GPU dry runs): stage through the host
        return gather_rows(local.cpu(), counts, group).to(local.device)
    words, mx = local.shape[1], max(max(counts), 1)
    if all(c == mx for c in counts):
        out = torch.empty((world * mx, words), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    pad = torch.zeros((mx, words), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    out = torch.empty((world * mx, words), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    return torch.cat([out[r * mx:r * mx + c] for r, c in enumerate(counts)], dim=0)


def gather_public(pub_local, n_total=None, group=None):
    """All-gather per-step public outputs.  pub_local: [n_local, words] int32 tensor (device or CPU).
    Returns [n_total, words] in global step order on every rank (shards = shard_range(n_total, r, world))."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return pub_local
    if n_total is None:
        cnt = torch.tensor([pub_local.shape[0]], dtype=torch.int64, device=pub_local.device)
        if pub_local.is_cuda and dist.get_backend(group) == "gloo":
            cnt = cnt.cpu()
        dist.all_reduce(cnt, group=group)
        n_total = int(cnt.item())
    return gather_rows(pub_local, [e - s for s, e in (shard_range(n_total, r, world) for r in range(world))], group)


def torch_allgather(device, group=None):
    """An all-gather for Comm.external(ctx, rank, nranks, fn) over torch.distributed: fn(d_send, d_recv, bytes_per_rank, stream)
    on raw device pointers.  "nccl" (= RCCL): the collective is enqueued behind `stream`'s work; "gloo" (CPU rehearsal of several
    ranks on one GPU): staged through the host, waits for `stream`."""
    class _Dev:
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (ptr, False), "version": 2, "strides": None}

    def fn(d_send, d_recv, nbytes, stream):
        world = dist.get_world_size(group)
        st = torch.cuda.ExternalStream(stream, device=device) if stream else torch.cuda.default_stream(device)
        with torch.cuda.stream(st):
            send = torch.as_tensor(_Dev(d_send, nbytes), device=device)
            recv = torch.as_tensor(_Dev(d_recv, nbytes * world), device=device)
            if dist.get_backend(group) == "gloo":
                host = torch.empty(nbytes * world, dtype=torch.uint8)
                dist.all_gather_into_tensor(host, send.cpu(), group=group)
                recv.copy_(host)
                st.synchronize()
            else:
                dist.all_gather_into_tensor(recv, send, group=group)
    return fn


class PublicExchange:
    """Pipelined exchange of the per-step public outputs: the all-gather of step i runs on RCCL's stream while the
    witness kernel of step i+1 runs on the compute stream.  `depth` public-output buffers alternate; a buffer is
    handed out again only after the gather that reads it has finished (stream-level wait, the host never blocks).

        ex = PublicExchange(n_local, words, device)
        for step in ...:
            pub = ex.next_buffer()            # int32 [n_local, words], to be written by the kernel of this step
            launch_kernel(..., pub.data_ptr())
            ex.post()                         # enqueue the all-gather of this step's outputs
        allpub = ex.finish()                  # [world * n_local, words] of the last step, global step order

    Shards must be equal (weak scaling: every rank runs n_local units)."""

    def __init__(self, n_local, words, device, depth=2, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = dist.is_initialized() and (self.world > 1 or dist.get_backend(group) == "nccl")
        self.staged = self.active and dist.get_backend(group) == "gloo"      # CPU rehearsal: through the host, synchronous
        self.bufs = [torch.zeros((n_local, words), dtype=torch.int32, device=device) for _ in range(depth)]
        self.outs = [torch.empty((self.world * n_local, words), dtype=torch.int32, device=device) for _ in range(depth)] \
            if self.active else self.bufs
        self.work = [None] * depth
        self.i = 0
        self.last = None

    def next_buffer(self):
        k = self.i % len(self.bufs)
        if self.work[k] is not None:
            self.work[k].wait()               # the current stream waits for that gather; no host sync
            self.work[k] = None
        return self.bufs[k]

    def post(self):
        k = self.i % len(self.bufs)
        if self.staged:
            host = torch.empty(self.outs[k].shape, dtype=torch.int32)
            dist.all_gather_into_tensor(host, self.bufs[k].cpu(), group=self.group)
            self.outs[k].copy_(host)
        elif self.active:
            self.work[k] = dist.all_gather_into_tensor(self.outs[k], self.bufs[k], group=self.group, async_op=True)
        self.last = k
        self.i += 1

    def finish(self):
        for k, w in enumerate(self.work):
            if w is not None:
                w.wait()
                self.work[k] = None
        return None if self.last is None else self.outs[self.last]
