"""Chained ("nova fold") mode driver: preimage -> step records -> step witnesses, streamed.

BASELINE configs 4/5: a preimage is split into 1 KiB chunks, each chunk into <= 16 blocks; every block is
one nova step (leaf steps), and every chunk path adds log2(n_chunks) parent steps.  The reference
does this for ONE chunk path, one step at a time (rust_fold/src/main.rs:41-203); here all steps of all
chunks are independent after the planner's native BLAKE3 pre-pass.

Pipeline per GPU (one process per GPU; ranks take contiguous chunk ranges):
  copy stream    : pinned host slices of the preimage -> HBM (64 B per step: negligible next to the
                   745 KB of witness each step writes, but overlapped anyway)
  compute stream : plan leaf steps of the slice -> nova witness kernel over batches of steps, bodies
                   written into a RING of batch buffers (a 1 GiB preimage is 27 TB of witness: bodies
                   are handed to a consumer per batch and then overwritten)
  exchange       : all-gather of chunk chaining values (32 B per chunk) so that every rank can build
                   the upper tree levels redundantly; the per-step public outputs stay per rank
                   unless the caller gathers them (sharding.gather_public)
"""
import numpy as np
import torch
import torch.distributed as dist

from . import B3W_OK, B3WError, lib
from .sharding import shard_range


def _chk(ctx, rc, what):
    if rc != B3W_OK:
        raise B3WError(rc, f"{what}: status {rc}: {ctx.last_error()}")


def fold_witnesses(ctx, preimage, batch_steps=16384, ring=2, slice_chunks=1024, with_parents=True, consumer=None,
                   device=None):
    """preimage: 1-D uint8 numpy array / torch CPU tensor (the whole preimage; every rank passes the same).
    consumer(bodies_view [k, body_bytes] uint8 CUDA, first_local_step, k): called after each batch is enqueued;
    it must enqueue its work on the current stream (the view is overwritten `ring` batches later).
    Returns dict(public=[n_local_steps, 15] int32 CUDA, status=[n_local_steps] int32 CUDA, root=[8] int32,
    n_leaf_steps, n_parent_steps, first_chunk, n_chunks_local, n_chunks)."""
    L = lib()
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    dev = device or torch.device("cuda", torch.cuda.current_device())
    host = torch.from_numpy(preimage) if isinstance(preimage, np.ndarray) else preimage
    assert host.dtype == torch.uint8 and host.dim() == 1 and host.numel() > 0
    ln = host.numel()
    n = L.b3w_chain_num_chunks(ln)
    c0, c1 = shard_range(n, rank, world)
    nl = c1 - c0
    P = L.b3w_chain_path_len(0, n)
    complete = (n & (n - 1)) == 0
    last_blocks = (max(ln - (n - 1) * 1024, 1) + 63) // 64
    has_last = c1 == n
    n_leaf = nl * 16 - ((16 - last_blocks) if has_last else 0)
    n_par = nl * P if (with_parents and complete) else 0
    body = ctx.body_bytes

    pinned = host if host.is_pinned() else None
    d_pre = torch.empty(max(nl, 1) * 1024, dtype=torch.uint8, device=dev)
    recs = torch.zeros((nl * 16 + n_par, 32), dtype=torch.int32, device=dev)
    cvs_local = torch.zeros((max(nl, 1), 8), dtype=torch.int32, device=dev)
    pub = torch.zeros((nl * 16 + n_par, 15), dtype=torch.int32, device=dev)
    status = torch.zeros((nl * 16 + n_par,), dtype=torch.int32, device=dev)
    # the ring of batch buffers: placed over two classes of HBM (b3w_bodies_alloc), allocated once per context
    cache = ctx.__dict__.setdefault("_ring_cache", {})
    key = (batch_steps, ring, dev.index)
    if key not in cache:
        for bufs in cache.values():
            for b in bufs:
                b.free()
        cache.clear()
        cache[key] = [ctx.alloc_bodies(batch_steps * body) for _ in range(ring)]
    bodies = [b.tensor()[: batch_steps * body].view(batch_steps, body) for b in cache[key]]
    compute = torch.cuda.current_stream(dev)
    copy = torch.cuda.Stream(dev)
    nbatch = 0

    def run_steps(first_row, count, first_step):
        """witness kernel over record rows [first_row, first_row+count), in ring-buffered batches"""
        nonlocal nbatch
        done = 0
        while done < count:
            k = min(batch_steps, count - done)
            slot = nbatch % ring
            r0 = first_row + done
            ctx.run_device(recs[r0:].data_ptr(), k, bodies[slot].data_ptr(), 0, pub[r0:].data_ptr(), status[r0:].data_ptr(),
                           compute.cuda_stream)
            if consumer is not None:
                consumer(bodies[slot][:k], first_step + done, k)
            nbatch += 1
            done += k

    # ---- leaf steps, slice by slice, H2D overlapped with planning + witness kernels of earlier slices
    lo = c0 * 1024
    for s0 in range(0, nl, slice_chunks):
        sc = min(slice_chunks, nl - s0)
        b0, b1 = lo + s0 * 1024, min(lo + (s0 + sc) * 1024, ln)
        ev = torch.cuda.Event()
        with torch.cuda.stream(copy):
            src = (pinned if pinned is not None else host)[b0:b1]
            d_pre[s0 * 1024: s0 * 1024 + (b1 - b0)].copy_(src, non_blocking=pinned is not None)
            ev.record(copy)
        compute.wait_event(ev)
        _chk(ctx, L.b3w_chain_plan_leaves_device(ctx.handle, d_pre[s0 * 1024:].data_ptr(), ln, c0 + s0, sc,
                                                 recs[s0 * 16:].data_ptr(), cvs_local[s0:].data_ptr(), compute.cuda_stream),
             "plan_leaves")
        steps_here = sc * 16 - ((16 - last_blocks) if (has_last and s0 + sc == nl) else 0)
        run_steps(s0 * 16, steps_here, s0 * 16)

    # ---- exchange: chunk chaining values of all ranks -> level 0 of the tree (32 B per chunk)
    levels = torch.zeros(((2 * n + 64) * 8,), dtype=torch.int32, device=dev)
    root = torch.zeros(8, dtype=torch.int32, device=dev)
    if world > 1:
        sizes = [shard_range(n, r, world) for r in range(world)]
        mx = max(e - s for s, e in sizes)
        pad = torch.zeros((mx, 8), dtype=torch.int32, device=dev)
        pad[:nl] = cvs_local[:nl]
        if dist.get_backend() == "gloo":              # CPU rehearsal of the exchange (tests): stage through the host
            allcv_h = torch.empty((world * mx, 8), dtype=torch.int32)
            dist.all_gather_into_tensor(allcv_h, pad.cpu())
            allcv = allcv_h.to(dev)
        else:                                         # RCCL over xGMI
            allcv = torch.empty((world * mx, 8), dtype=torch.int32, device=dev)
            dist.all_gather_into_tensor(allcv, pad)
        for r, (s, e) in enumerate(sizes):
            levels[s * 8: e * 8] = allcv[r * mx: r * mx + (e - s)].reshape(-1)
    else:
        levels[: n * 8] = cvs_local[:n].reshape(-1)
    _chk(ctx, L.b3w_chain_tree_device(ctx.handle, levels.data_ptr(), n, root.data_ptr(), compute.cuda_stream), "tree")

    # ---- parent steps of the local chunks (complete trees)
    if n_par:
        _chk(ctx, L.b3w_chain_plan_parents_device(ctx.handle, levels.data_ptr(), n, ln, c0, nl, recs[nl * 16:].data_ptr(),
                                                  compute.cuda_stream), "plan_parents")
        run_steps(nl * 16, n_par, n_leaf)

    # rows of a partial last chunk that hold no step are dropped from the returned views
    if has_last and last_blocks < 16:
        keep = torch.ones(recs.shape[0], dtype=torch.bool, device=dev)
        keep[(nl - 1) * 16 + last_blocks: nl * 16] = False
        recs, pub, status = recs[keep], pub[keep], status[keep]
    return dict(public=pub, status=status, records=recs, root=root, n_leaf_steps=n_leaf, n_parent_steps=n_par,
                first_chunk=c0, n_chunks_local=nl, n_chunks=n, path_len=P, batches=nbatch)
