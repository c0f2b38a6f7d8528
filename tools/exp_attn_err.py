"""bf16 first-step logits error vs the fp32 golden, flash (MFMA) vs simple attention in the prefill."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "index-tts-ipex_amd"))
from itts_hip import config as icfg, engine as ieng, synth
CFG = icfg.indextts_1_5()
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "full_decode_b1.npz"))
eng = ieng.build_engine(CFG, "bf16", parts=("gpt",))
mel = torch.from_numpy(synth.prompt_mel(511, seed=7))
cond = eng.conditioning(mel)
eng.prefill(cond, g["text"], 48)
codes, lg = eng.fetch(logits=True)
eng._exit()
idx, val = g["top_idx"][0], g["top_val"][0]
print("mode", os.environ.get("ITTS_ATTN_SIMPLE", "flash"), "top8 ref", np.round(val, 3), "got", np.round(lg[0, idx], 3), "maxdiff", np.abs(lg[0, idx] - val).max(), "argmax", lg[0].argmax(), idx[0])
