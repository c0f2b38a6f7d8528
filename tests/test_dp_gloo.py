"""CPU, world_size 2, gloo: the data-parallel plumbing around the hot path (no GPU): utterance partition, weight
arena broadcast (what RCCL does over xGMI on the GPU node) and the variable-length waveform gather."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from itts_hip import dp


def test_partition_covers_and_balances():
    lens = [50, 120, 30, 90, 75, 110, 20, 60, 100]
    for world in (1, 2, 4, 8):
        parts = [dp.partition(lens, world, r) for r in range(world)]
        flat = sorted(i for p in parts for i in p)
        assert flat == list(range(len(lens)))
        loads = [sum(lens[i] for i in p) for p in parts]
        if world == 2:
            assert max(loads) - min(loads) <= max(lens)
    assert dp.partition([], 4, 1) == []


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if rank == 0:
            arena = torch.arange(100000, dtype=torch.int64).view(torch.uint8).clone()
            manifest = [("a.weight", 0, 1, (10, 10)), ("b.bias", 256, 0, (7,))]
            buf, man = dp.broadcast_arena(arena, manifest)
        else:
            buf, man = dp.broadcast_arena(None, None)
        ok_arena = bool((buf.view(torch.int64) == torch.arange(100000)).all()) and man[1][0] == "b.bias"
        lens = [5, 9, 3, 7, 8]
        mine = dp.partition(lens, world, rank)
        local = {i: (np.arange(lens[i] * 10, dtype=np.int16) + i) for i in mine}  # stand-in for the synthesis
        full = dp.gather_waveforms(local, len(lens))
        if rank == 0:
            ok = ok_arena and all(full[i].shape[0] == lens[i] * 10 and full[i][0] == i for i in range(len(lens)))
            out.put(("ok" if ok else "bad", mine))
        else:
            out.put(("ok" if ok_arena else "bad", mine))
    finally:
        dist.destroy_process_group()


def test_broadcast_and_gather_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[0] == "ok" for r in res), res
    assert sorted(i for r in res for i in r[1]) == [0, 1, 2, 3, 4]
