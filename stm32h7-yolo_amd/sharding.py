"""Batch sharding across the GPUs of one node: frames are independent, so rank r takes the
contiguous slice [r*B/G, (r+1)*B/G) and the only exchange is one all-gather of the int8 heads (RCCL over xGMI
when the process group backend is "nccl"; "gloo" in the CPU tests)."""
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous split; the first n % world ranks take one extra frame."""
    base, extra = divmod(n, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def all_gather_heads(local_heads, n_total, group=None):
    """local_heads: tensor [n_local, ...] on this rank's device (int8 heads [n,7,7,18], or any per-frame record
    array such as detection records [n, cap, 28] / counts [n]) -> [n_total, ...] on every rank.
    Uneven shards are padded to the largest shard for the fixed-shape collective and trimmed afterwards."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local_heads
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    cap = max(b - a for a, b in sizes)
    pad = torch.zeros((cap,) + tuple(local_heads.shape[1:]), dtype=local_heads.dtype, device=local_heads.device)
    pad[: local_heads.shape[0]] = local_heads
    out = torch.empty((world * cap,) + tuple(local_heads.shape[1:]), dtype=local_heads.dtype, device=local_heads.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    if all(b - a == cap for a, b in sizes):
        return out
    return torch.cat([out[r * cap: r * cap + (b - a)] for r, (a, b) in enumerate(sizes)], dim=0)


all_gather_rows = all_gather_heads      # same collective for per-frame detection records and counts
