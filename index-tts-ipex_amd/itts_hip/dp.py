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
    elif dist.get_backend() != "nccl" and buf.is_cuda:
        buf = buf.cpu()
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


def world() -> tuple:
    """(rank, world_size) of the current process group, (0, 1) outside torch.distributed."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def replicate_packed(engine, packers: Sequence, src: int = 0):
    """Load the same packed checkpoint parts into `engine` on every rank: rank `src` runs each packer (a callable
    returning the {name: (tag, ndarray)} dict of itts_hip.pack) and uploads it into a WeightArena, the other ranks
    receive the arena bytes with ONE broadcast per part (RCCL over xGMI with the nccl backend; staged through the host
    with gloo, which is how several ranks rehearse on a one-GPU box) and bind them.  Returns the arenas."""
    from . import engine as ieng

    rank, ws = world()
    arenas = []
    for fn in packers:
        if ws == 1:
            arenas.append(engine.load_packed(fn()))
            continue
        half = getattr(engine, "half_dtype", torch.bfloat16)  # the engine's 16-bit storage type (float16 for the f16 build)
        arena = ieng.WeightArena(fn(), engine.dt, engine.device, half) if rank == src else None
        on_host = dist.get_backend() != "nccl"
        buf = None if arena is None else (arena.buf.cpu() if on_host else arena.buf)
        buf, manifest = broadcast_arena(buf, None if arena is None else arena.manifest, src=src)
        if arena is None:
            arena = ieng.WeightArena.__new__(ieng.WeightArena)
            arena.dtype, arena.manifest, arena.nbytes, arena.half = engine.dt, manifest, int(buf.numel()), half
            arena.buf = buf.to(engine.device) if on_host else buf
        engine.load_packed(None, arena=arena)
        arenas.append(arena)
    return arenas


def run_sharded(lengths: Sequence[int], synth_fn, gather: bool = True, dst: int = 0):
    """Data-parallel synthesis of len(lengths) independent utterances: this rank's shard (`partition`) goes through
    `synth_fn(indices) -> {index: int16 waveform}` in ONE call (so the rank can batch it), no collective on the data
    path; with `gather` the waveforms are collected on `dst` in original order (None on the other ranks).
    Returns (waveforms or None, my_indices)."""
    rank, ws = world()
    mine = partition(lengths, ws, rank)
    local = synth_fn(mine) if mine else {}
    assert set(local) == set(mine), (sorted(local), mine)
    if ws == 1:
        return [local[i] for i in range(len(lengths))], mine
    if not gather:
        return None, mine
    return gather_waveforms(local, len(lengths), dst=dst), mine


def synthesize_sharded(tts, prompt_mels, texts, gather: bool = True, **kw):
    """Product entry point for multi-GPU batches: every rank holds an `IndexTTS` on its own GPU and calls this with the
    SAME arguments; each synthesises its shard with `IndexTTS.infer_batch` and rank 0 gets all (24000, int16) results."""
    import torch as _t

    shared = isinstance(prompt_mels, _t.Tensor)
    lens = [len(t) if not isinstance(t, str) else len(t.encode("utf-8")) for t in texts]
    if not isinstance(texts[0], str):
        lens = [sum(len(s) for s in t) for t in texts]

    def fn(idx):
        res = tts.infer_batch(prompt_mels if shared else [prompt_mels[i] for i in idx], [texts[i] for i in idx], **kw)
        return {i: r[1] for i, r in zip(idx, res)}

    return run_sharded(lens, fn, gather=gather)
