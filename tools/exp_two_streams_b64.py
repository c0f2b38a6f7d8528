"""Experiment: the 64-row decode batch (BASELINE config 3) as ONE stream vs two concurrent 32-row streams vs four 16-row
streams (separate engines sharing the weight arena, one HIP stream each): do the launch-bound projections of one half
overlap the HBM-bound cache attention of the other?"""
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
T = int(os.environ.get("T", "200"))
NR = 64
texts = np.stack([synth.text_ids(105, 11 + i, 12000) for i in range(NR)]).astype(np.int32)

def run(groups, chunk):
    for e, rows in groups:
        e.prefill(cond, texts[rows], T, 10.0, True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(0, T - 1, chunk):
        n = min(chunk, T - 1 - k)
        for e, rows in groups:
            e.decode(n)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for e, _ in groups:
        e._exit()
    return dt

for name, groups in [("1 stream  x 64 rows", [(engs[0], list(range(64)))]),
                     ("2 streams x 32 rows", [(engs[i], list(range(32 * i, 32 * i + 32))) for i in range(2)]),
                     ("4 streams x 16 rows", [(engs[i], list(range(16 * i, 16 * i + 16))) for i in range(4)]),
                     ("1 stream  x 32 rows", [(engs[0], list(range(32)))])]:
    for chunk in (8, 1):
        run(groups, chunk)
        dt = min(run(groups, chunk) for _ in range(2))
        rows = sum(len(r) for _, r in groups)
        print(f"{name} (decode({chunk}) calls): {dt * 1e3:7.1f} ms for {T - 1} steps -> {dt / (T - 1) * 1e3:.3f} ms/step, {rows * (T - 1) / dt:8.0f} tokens/s", flush=True)
