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


class _FakeEngine:
    """Stand-in for itts_hip.engine.Engine on the CPU: records the arenas replicate_packed binds."""

    def __init__(self):
        from itts_hip import lib as L

        self.dt, self.device, self.bound = L.BF16, torch.device("cpu"), []

    def load_packed(self, packed, arena=None):
        from itts_hip import engine as ieng

        a = arena or ieng.WeightArena(packed, self.dt, self.device)
        self.bound.append(a)
        return a


def _worker_sharded(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        calls = []

        def packer(tag):
            def fn():
                calls.append(tag)  # only the source rank may materialise a checkpoint
                rng = np.random.default_rng(7)
                return {f"{tag}.w": ("w", rng.standard_normal((33, 16)).astype(np.float32)),
                        f"{tag}.b": ("f", rng.standard_normal((33,)).astype(np.float32))}
            return fn

        eng = _FakeEngine()
        arenas = dp.replicate_packed(eng, [packer("gpt"), packer("bigvgan")])
        digest = [int(a.buf.to(torch.int64).sum()) for a in arenas]
        names = [[m[0] for m in a.manifest] for a in arenas]
        w = eng.bound[0].view("gpt.w").float()
        lens = [7, 3, 9, 5, 8, 2]

        def synth_fn(idx):  # one call per rank with its whole shard (so the rank can batch it)
            return {i: (np.full(lens[i] * 4, i, dtype=np.int16)) for i in idx}

        full, mine = dp.run_sharded(lens, synth_fn)
        ok = (rank == 0) == (full is not None)
        if full is not None:
            ok = ok and all(f.shape[0] == lens[i] * 4 and int(f[0]) == i for i, f in enumerate(full))
        out.put((ok, calls, digest, names, float(w.abs().sum()), mine))
    finally:
        dist.destroy_process_group()


def test_replicate_and_run_sharded_world2():
    """The path bench.py --gpus N and dp.synthesize_sharded use: weights replicated by broadcast from rank 0 (the other
    rank never packs a checkpoint), shards dealt by partition, one synthesis call per rank, results gathered in order."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_sharded, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[0] for r in res), res
    assert sorted(len(r[1]) for r in res) == [0, 2]  # packers ran on rank 0 only
    assert res[0][2] == res[1][2] and res[0][3] == res[1][3] and res[0][4] == res[1][4]  # identical arenas everywhere
    assert sorted(i for r in res for i in r[5]) == list(range(6))


def test_bench_launcher_refuses_mismatch():
    """`bench.py --gpus N` under a torchrun environment of a different size must fail loudly, not run 1 GPU silently."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--micro"], env=env, capture_output=True, text=True)
    assert r.returncode != 0 and "does not match WORLD_SIZE" in (r.stderr + r.stdout)
