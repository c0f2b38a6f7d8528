"""Data-parallel sharding of utterances over the GPUs of one node.

Utterances are independent, so the data path has NO collective: one process per GPU, each rank synthesises its own
shard.  The only communication is (1) a one-off broadcast of the packed weight arena from rank 0 (RCCL over xGMI on
GPUs, gloo on CPU in tests) and (2) the gather of variable-length waveforms to rank 0 at the end."""
from __future__ import annotations

from typing import List, Sequence

import numpy as np
import torch
import torch.distributed as dist


def partition(lengths: Sequence[int], world: int, rank: int) -> List[int]:
    """Indices of this rank's items: sort by descending length, deal round-robin in serpentine order (balances the
    number of AR steps per rank), return in ascending original index."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    mine = []
    for pos, idx in enumerate(order):
        rnd, slot = divmod(pos, world)
        owner = slot if rnd % 2 == 0 else world - 1 - slot
        if owner == rank:
            mine.append(idx)
    return sorted(mine)


def broadcast_arena(buf: torch.Tensor, manifest, src: int = 0):
    """Broadcast the manifest (python object) and the arena bytes.  Non-source ranks pass buf=None/manifest=None and
    receive freshly allocated ones on `device` of the source's choosing (same device type as their default)."""
    meta = [manifest, None if buf is None else int(buf.numel())]
    dist.broadcast_object_list(meta, src=src)
    manifest, n = meta
    if dist.get_rank() != src:
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
        buf = torch.empty(n, dtype=torch.uint8, device=dev)
    dist.broadcast(buf, src=src)
    return buf, manifest


def gather_waveforms(local: dict, n_total: int, dst: int = 0):
    """local: {utterance index: int16 numpy waveform}.  Returns the full ordered list on `dst`, None elsewhere."""
    world, rank = dist.get_world_size(), dist.get_rank()
    gathered = [None] * world if rank == dst else None
    dist.gather_object({int(k): np.asarray(v) for k, v in local.items()}, gathered, dst=dst)
    if rank != dst:
        return None
    merged = {}
    for d in gathered:
        merged.update(d)
    assert len(merged) == n_total, (len(merged), n_total)
    return [merged[i] for i in range(n_total)]
