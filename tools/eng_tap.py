"""Debugging aid: the edges of ONE block (ITTS_TAP_LAYER) of the first decode step, persistent engine vs launch path."""
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
eng = ieng.build_engine(CFG, "bf16", parts=("gpt",))
cond = eng.conditioning(torch.from_numpy(synth.prompt_mel(511, seed=7)))
text = np.stack([synth.text_ids(105, 11 + r, CFG.gpt.number_text_tokens) for r in range(rows)]).astype(np.int32)


def run(no_engine, names):
    eng.debug(taps=True, no_engine=no_engine, engine=not no_engine, no_graph=True)
    eng.prefill(cond, text, 64, 10.0, True)
    eng.decode(1)
    eng.fetch()
    eng._exit()
    out = {n: eng.fetch_tap(n).copy() for n in names}
    eng.debug()
    return out


lp = run(True, ["lp_qkv", "lp_h1", "lp_act", "lp_h2"])
en = run(False, ["eng_qkv", "eng_h1", "eng_act", "eng_h2"])
for k in ("qkv", "h1", "act", "h2"):
    a, b = en["eng_" + k], lp["lp_" + k]
    bad = np.nonzero(a.view(np.uint32) != b.view(np.uint32))[0]
    print(f"{k}: n={a.size} mismatching {bad.size} max|d| {np.abs(a - b).max():.3e}", "first:", [(int(i), float(a[i]), float(b[i])) for i in bad[:6]])
