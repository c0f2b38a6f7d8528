"""Debugging aid for the persistent decode engine: one decode step after the same prefill, engine vs launch path,
bitwise and by magnitude.  ITTS_ENGINE_LAYERS=n runs blocks [0, n) on the engine and the rest as launches."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "index-tts-ipex_amd")):
    sys.path.insert(0, p)
from itts_hip import config as icfg, engine as ieng, synth  # noqa: E402

CFG = icfg.indextts_1_5()
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
eng = ieng.build_engine(CFG, "bf16", parts=("gpt",))
cond = eng.conditioning(torch.from_numpy(synth.prompt_mel(511, seed=7)))
text = np.stack([synth.text_ids(105, 11 + r, CFG.gpt.number_text_tokens) for r in range(rows)]).astype(np.int32)


def run(no_engine):
    eng.debug(no_engine=no_engine, engine=not no_engine, no_graph=True)
    eng.prefill(cond, text, 64, 10.0, True)
    out = []
    for _ in range(steps):
        eng.decode(1)
        codes, lg = eng.fetch(logits=True)
        out.append((codes.copy(), lg.copy()))
    eng._exit()
    eng.debug()
    return out


ref = run(True)
got = run(False)
got2 = run(False)
for k in range(steps):
    d = np.abs(got[k][1] - ref[k][1])
    d2 = np.abs(got[k][1] - got2[k][1])
    print(f"step {k}: max|dlogits| {d.max():.3e} (rel {d.max() / np.abs(ref[k][1]).max():.3e}), bit-equal {np.array_equal(got[k][1].view(np.uint32), ref[k][1].view(np.uint32))}, "
          f"engine run-to-run max diff {d2.max():.3e}, codes equal {np.array_equal(got[k][0][:, :k + 2], ref[k][0][:, :k + 2])}")
