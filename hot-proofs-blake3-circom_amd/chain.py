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
  exchange       : (1) all-gather of chunk chaining values (32 B per chunk) so that every rank can build
                   the upper tree levels redundantly; (2) all-gather of every step's h_out (8 u32 per
                   step: 16 384 x 8 for the 1 MiB preimage of BASELINE config 4) — the running chaining
                   value the fold consumes as z_{i+1} (Blake3CompressPubIO::to_vec,
                   rust_fold/src/blake3_circuit.rs:111-123; fed back at rust_fold/src/main.rs:166-179) —
                   into global step order on every rank.  RCCL over xGMI through torch.distributed
                   ("nccl"), or natively through a b3w_comm (b3w_chain_allgather_hout, include/b3wit.h)
"""
import ctypes

import numpy as np
import torch
import torch.distributed as dist

from . import B3W_OK, PLACEMENT_NAMES, B3WError, lib
from .sharding import gather_rows, shard_range


def _chk(ctx, rc, what):
    if rc != B3W_OK:
        raise B3WError(rc, f"{what}: status {rc}: {ctx.last_error()}")


class _DevArray:
    """zero-copy torch view of library-owned device memory (lives as long as the b3w_chain object)"""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (ptr, False), "version": 2,
                                         "strides": None}


def _view(ptr, shape, typestr, dev):
    return torch.as_tensor(_DevArray(ptr, shape, typestr), device=dev)


_CONSUMER = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32,
                             ctypes.c_void_p)


def step_shards(preimage_len, world, with_parents=True):
    """Per rank: (first chunk, chunks, leaf steps, first parent row, parent steps) of the contiguous chunk shards — the geometry
    of the h_out exchange (rank order = global step order)."""
    L = lib()
    n = L.b3w_chain_num_chunks(preimage_len)
    last_bytes = preimage_len - (n - 1) * 1024 if preimage_len > (n - 1) * 1024 else 0
    last_blocks = -(-last_bytes // 64) if last_bytes else 1
    out = []
    for r in range(world):
        f, e = shard_range(n, r, world)
        k = e - f
        leaf = k * 16 - ((16 - last_blocks) if (k and e == n) else 0)
        p0 = L.b3w_chain_parent_row(f, n) if with_parents else 0
        p1 = L.b3w_chain_parent_row(e, n) if with_parents else 0
        out.append((f, k, leaf, p0, p1 - p0))
    return out


def gather_h_out(public_local, n_leaf_local, preimage_len, with_parents=True, group=None):
    """The fold's exchange: all-gather of h_out (public words 2..9) of every step of a sharded pass.  public_local: this rank's
    [n_leaf_local + n_parent_local, 15] public outputs (leaf steps first).  Returns (leaf [n_leaf_total, 8], parents
    [n_parent_total, 8]) in global step order — (chunk, block) and (chunk, height) — on every rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    h = public_local[:, 2:10]
    if world == 1:
        return h[:n_leaf_local], h[n_leaf_local:]
    sh = step_shards(preimage_len, world, with_parents)
    leaf = gather_rows(h[:n_leaf_local].contiguous(), [x[2] for x in sh], group)
    par = gather_rows(h[n_leaf_local:].contiguous(), [x[4] for x in sh], group)
    return leaf, par


COMMIT_OVERLAP = {"auto": -1, "serial": 0, "free": 1, "gated": 2}      # b3w_chain_commit_overlap (include/b3wit.h)


def fold_witnesses(ctx, preimage, batch_steps=16384, ring=2, slice_chunks=1024, with_parents=True, consumer=None,
                   device=None, commit_only=None, gather_hout=True, comm=None, commit_records=None, check=None, commit_overlap="auto"):
    """preimage: 1-D uint8 numpy array / torch CPU tensor (the whole preimage; every rank passes the same).
    consumer(bodies_view [k, body_bytes] uint8 CUDA, first_local_step, k): called after each batch is enqueued;
    it must enqueue its work on the current stream (the view is overwritten `ring` batches later).
    commit_only=(CommitKey, d_points): no bodies at all — one commitment per step, computed from the step records
    (b3w_chain_commit_only), into the caller's [n_steps, 64] uint8 CUDA tensor.
    commit_records=(CommitKey, d_points): the same commitments from the records WHILE the bodies are written and handed to the
    consumer as usual (b3w_chain_commit_from_records): the fold-shaped pass, without reading the bodies back for the commitment.
    commit_overlap: where those commitments run — "serial" (caller's stream), "free" (the chain's commit stream, beside the witness
    kernels of this and later batches), "gated" (beside the batch's own witness kernel only; check and consumer start when both
    are done), "auto" (gated when something reads the batch, else free): b3w_chain_commit_overlap.
    check=R1cs: every batch of step witnesses is checked against the step circuit's constraints while it sits in the ring, before
    the consumer sees it (b3w_chain_check_constraints); the result gains violations=[n_local_steps] int32 CUDA (0 = satisfied).
    gather_hout: all-gather every step's h_out across the ranks inside the pass (the fold's exchange, module docstring);
    comm: a native b3w_comm handle (b3w_comm_create) — the two exchanges then go through the library's own RCCL calls
    (b3w_chain_run_parents_sharded, b3w_chain_allgather_hout) instead of torch.distributed.
    Returns dict(public=[n_local_steps, 15] int32 CUDA, status=[n_local_steps] int32 CUDA, root=[8] int32,
    h_out_all=[n_leaf_steps of ALL ranks, 8] int32, h_out_parents_all=[n_parent_steps of all ranks, 8] (global step order;
    None with gather_hout=False on several ranks), n_leaf_steps, n_parent_steps, first_chunk, n_chunks_local, n_chunks).
    The device arrays belong to a b3w_chain object cached on `ctx` and are overwritten by the next fold of the same shape.

    The pass itself is the library's native driver (b3w_chain_run_leaves / b3w_chain_run_parents, include/b3wit.h);
    this function adds what needs the process group: the all-gathers of the chunk chaining values and of h_out."""
    L = lib()
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if comm is not None:
        world, rank = comm.nranks, comm.rank
    dev = device or torch.device("cuda", torch.cuda.current_device())
    host = torch.from_numpy(preimage) if isinstance(preimage, np.ndarray) else preimage
    assert host.dtype == torch.uint8 and host.dim() == 1 and host.numel() > 0 and host.is_contiguous()
    ln = host.numel()
    n = L.b3w_chain_num_chunks(ln)
    c0, c1 = shard_range(n, rank, world)
    nl = c1 - c0
    body = ctx.body_bytes

    cache = ctx.__dict__.setdefault("_chain_cache", {})
    key = (ln, c0, nl, batch_steps, ring, bool(with_parents))
    if key not in cache:
        for h in cache.values():
            L.b3w_chain_destroy(h)
        cache.clear()
        h = ctypes.c_void_p()
        _chk(ctx, L.b3w_chain_create(ctx.handle, ln, c0, nl, batch_steps, ring, 1 if with_parents else 0, ctypes.byref(h)),
             "b3w_chain_create")
        cache[key] = h
    h = cache[key]
    nleaf, npar, nch, P, pl = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint32(), ctypes.c_int32()
    L.b3w_chain_info(h, ctypes.byref(nleaf), ctypes.byref(npar), ctypes.byref(nch), ctypes.byref(P), ctypes.byref(pl))
    n_leaf, n_par = nleaf.value, npar.value
    if commit_only is not None:
        _chk(ctx, L.b3w_chain_commit_only(h, commit_only[0].handle, commit_only[1].data_ptr()), "b3w_chain_commit_only")
    elif commit_records is not None:
        _chk(ctx, L.b3w_chain_commit_from_records(h, commit_records[0].handle, commit_records[1].data_ptr()), "b3w_chain_commit_from_records")
    else:
        _chk(ctx, L.b3w_chain_commit_only(h, None, None), "b3w_chain_commit_only")
    if commit_records is not None:
        _chk(ctx, L.b3w_chain_commit_overlap(h, COMMIT_OVERLAP[commit_overlap]), "b3w_chain_commit_overlap")
    _chk(ctx, L.b3w_chain_check_constraints(h, check.handle if check is not None else None), "b3w_chain_check_constraints")
    compute = torch.cuda.current_stream(dev)
    nbatch = [0]
    cb = _CONSUMER()
    if consumer is not None:
        def _cb(user, d_bodies, pitch, first_step, count, stream):
            nbatch[0] += 1
            consumer(_view(d_bodies, (count, body), "|u1", dev), first_step, count)
        cb = _CONSUMER(_cb)
    _chk(ctx, L.b3w_chain_run_leaves(h, host.data_ptr(), cb, None, compute.cuda_stream), "b3w_chain_run_leaves")

    rows = n_leaf + n_par
    public = _view(L.b3w_chain_public(h), (rows, 15), "<i4", dev)
    h_leaf, h_par = None, None
    timer = None                                   # exchange_ms(out) reads it: how long the two exchanges took on this rank's device
    if comm is not None:
        # ---- both exchanges natively, through the b3w_comm (RCCL, host shared memory or the caller's collective)
        _chk(ctx, L.b3w_chain_run_parents_sharded(h, comm.handle, cb, None, compute.cuda_stream), "b3w_chain_run_parents_sharded")
        if gather_hout:
            sh = step_shards(ln, world, with_parents)
            h_leaf = torch.empty((sum(x[2] for x in sh), 8), dtype=torch.int32, device=dev)
            h_par = torch.empty((sum(x[4] for x in sh), 8), dtype=torch.int32, device=dev)
            with torch.cuda.stream(compute):
                _chk(ctx, L.b3w_chain_allgather_hout(h, comm.handle, h_leaf.data_ptr(), h_par.data_ptr() if h_par.numel() else None,
                                                     compute.cuda_stream), "b3w_chain_allgather_hout")
        timer = ("native", h, bool(gather_hout))
    else:
        # ---- exchange 1: chunk chaining values of all ranks (32 B per chunk)
        all_cvs = None
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if world > 1 else None
        if world > 1:
            ev[0].record(compute)
            cvs_local = _view(L.b3w_chain_local_cvs(h), (max(nl, 1), 8), "<i4", dev)
            all_cvs = gather_rows(cvs_local[:nl], [e - s for s, e in (shard_range(n, r, world) for r in range(world))]).contiguous()
            ev[1].record(compute)
        _chk(ctx, L.b3w_chain_run_parents(h, all_cvs.data_ptr() if all_cvs is not None else None, cb, None, compute.cuda_stream),
             "b3w_chain_run_parents")
        # ---- exchange 2: h_out of every step (8 u32 per step), global step order on every rank
        if gather_hout or world == 1:
            if world > 1:
                ev[2].record(compute)
            h_leaf, h_par = gather_h_out(public, n_leaf, ln, with_parents)
            if world > 1:
                ev[3].record(compute)
        if world > 1:
            timer = ("torch", ev, bool(gather_hout))

    return dict(public=public, h_out_all=h_leaf, h_out_parents_all=h_par, chunk_cvs_local=_view(L.b3w_chain_local_cvs(h), (max(nl, 1), 8), "<i4", dev)[:nl],
                status=_view(L.b3w_chain_status(h), (rows,), "<i4", dev),
                violations=_view(L.b3w_chain_violations_device(h), (rows,), "<i4", dev) if check is not None else None,
                records=_view(L.b3w_chain_records(h), (rows, 32), "<i4", dev), root=_view(L.b3w_chain_root(h), (8,), "<i4", dev),
                n_leaf_steps=n_leaf, n_parent_steps=n_par, first_chunk=c0, n_chunks_local=nl, n_chunks=n, path_len=P.value,
                batches=-(-n_leaf // batch_steps) + -(-n_par // batch_steps) if consumer is None else nbatch[0],
                placement=PLACEMENT_NAMES.get(pl.value, "plain"), exchange_timer=timer)


def exchange_ms(out):
    """(chunk-CV exchange ms, h_out exchange ms) of the pass fold_witnesses returned `out` for, on this rank's device (HIP events
    on the compute stream around staging + collective + scatter); waits for the events.  (0, 0) on one rank."""
    t = out.get("exchange_timer")
    if t is None:
        return 0.0, 0.0
    if t[0] == "native":
        ms = (ctypes.c_float * 2)()
        lib().b3w_chain_exchange_ms(t[1], ms)
        return float(ms[0]), float(ms[1]) if t[2] else 0.0
    ev = t[1]
    ev[1].synchronize()
    a = ev[0].elapsed_time(ev[1])
    b = 0.0
    if t[2]:
        ev[3].synchronize()
        b = ev[2].elapsed_time(ev[3])
    return a, b
