"""Experiment: one B=2 decode stream vs two concurrent B=1 streams (two engines, shared nothing) on one GPU."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "index-tts-ipex_amd"))
from itts_hip import config, engine, synth, pack

cfg = config.indextts_1_5()
packed = pack.pack_gpt(synth.gpt_state_dict(cfg, 1234), cfg)
engs = []
for i in range(4):
    e = engine.Engine(cfg, "bf16", "cuda:0")
    e.load_packed(packed, arena=engs[0].arenas[0] if engs else None)
    e.finalize()
    engs.append(e)
mel = torch.from_numpy(synth.prompt_mel(511, seed=7)).cuda()
cond = engs[0].conditioning(mel)
T = 240
texts = np.stack([synth.text_ids(105, 11 + i, 12000) for i in range(4)]).astype(np.int32)

def run(groups):
    """groups: list of (engine, rows) decoded concurrently"""
    for e, rows in groups:
        e.prefill(cond, texts[rows], T, 10.0, True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(0, T - 1, 8):
        n = min(8, T - 1 - k)
        for e, rows in groups:
            e.decode(n)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for e, _ in groups:
        e._exit()
    return dt

for name, groups in [("1 stream  x B=2", [(engs[0], [0, 1])]),
                     ("2 streams x B=1", [(engs[0], [0]), (engs[1], [1])]),
                     ("1 stream  x B=4", [(engs[0], [0, 1, 2, 3])]),
                     ("2 streams x B=2", [(engs[0], [0, 1]), (engs[1], [2, 3])]),
                     ("4 streams x B=1", [(engs[i], [i]) for i in range(4)])]:
    run(groups)
    dt = min(run(groups) for _ in range(3))
    rows = sum(len(r) for _, r in groups)
    print(f"{name}: {dt * 1e3:7.1f} ms for {T - 1} steps -> {dt / (T - 1) * 1e3:.3f} ms/step, {rows * (T - 1) / dt:8.0f} tokens/s")
