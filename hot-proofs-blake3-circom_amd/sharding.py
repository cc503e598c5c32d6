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


def gather_public(pub_local, n_total=None, group=None):
    """All-gather per-step public outputs.  pub_local: [n_local, words] int32 tensor (device or CPU).
    Returns [n_total, words] in global step order on every rank.  Ragged shards are padded to the
    largest shard for the collective and trimmed afterwards."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return pub_local
    words = pub_local.shape[1]
    if pub_local.is_cuda and dist.get_backend(group) == "gloo":
        # CPU rehearsal of the exchange (tests / single-GPU dry runs): stage through the host
        return gather_public(pub_local.cpu(), n_total, group).to(pub_local.device)
    if n_total is None:
        cnt = torch.tensor([pub_local.shape[0]], dtype=torch.int64, device=pub_local.device)
        dist.all_reduce(cnt, group=group)
        n_total = int(cnt.item())
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    mx = max(e - s for s, e in sizes)
    if all(e - s == mx for s, e in sizes):
        out = torch.empty((world * mx, words), dtype=pub_local.dtype, device=pub_local.device)
        dist.all_gather_into_tensor(out, pub_local.contiguous(), group=group)
        return out
    pad = torch.zeros((mx, words), dtype=pub_local.dtype, device=pub_local.device)
    pad[:pub_local.shape[0]] = pub_local
    out = torch.empty((world * mx, words), dtype=pub_local.dtype, device=pub_local.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    return torch.cat([out[r * mx:r * mx + (e - s)] for r, (s, e) in enumerate(sizes)], dim=0)
