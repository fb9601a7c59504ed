"""Batch sharding of independent image+prompt samples over the GPUs of one node.

The hot path has no cross-sample dependency: every rank holds a full replica of the W4
weights (7B: 3.7 GB, 72B: ~36 GB; both fit one 288 GB MI355X) and prefills its own samples.
The one exchange step is an all-gather of last-token logits (B_local x vocab, fp16: 304 KB per
Qwen2-VL sample) -- latency bound, so a single RCCL all_gather per batch is the right shape.
Sample i goes to rank i % world, the striding the reference's vendored eval harness uses
(third/VLMEvalKit/vlmeval/inference.py:89-90).
"""
from __future__ import annotations

from typing import List, Sequence

import torch
import torch.distributed as dist


def shard_indices(n_samples: int, rank: int, world: int) -> List[int]:
    """Indices of the samples rank ``rank`` processes (i % world == rank)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of size {world}")
    return list(range(rank, n_samples, world))


def padded_local_count(n_samples: int, world: int) -> int:
    """Samples per rank after padding the batch to a multiple of ``world``."""
    return (n_samples + world - 1) // world


def gather_logits(local: torch.Tensor, n_samples: int, group=None) -> torch.Tensor:
    """All-gather per-rank last-token logits [B_local, V] and restore sample order.

    Ranks with fewer real samples than ``padded_local_count`` pad with zeros; the padding is
    dropped after the gather.  Returns [n_samples, V] on every rank."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    per = padded_local_count(n_samples, world)
    if local.shape[0] > per:
        raise ValueError("more local samples than the padded per-rank count")
    buf = local
    if local.shape[0] < per:
        pad = torch.zeros((per - local.shape[0], local.shape[1]), dtype=local.dtype, device=local.device)
        buf = torch.cat((local, pad), dim=0)
    out = torch.empty((world * per, local.shape[1]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, buf.contiguous(), group=group)
    # out[r * per + j] is sample r + j * world
    order = [r * per + j for i in range(n_samples) for r, j in [(i % world, i // world)]]
    assert len(shard_indices(n_samples, rank, world)) == local.shape[0]
    return out[torch.tensor(order, device=out.device)]


def broadcast_scales(scales: Sequence[float], src: int = 0, group=None, device=None) -> List[float]:
    """Static activation scales are replicated constants: calibrate on ``src``, broadcast."""
    t = torch.tensor(list(scales), dtype=torch.float64, device=device)
    dist.broadcast(t, src=src, group=group)
    return t.tolist()
