"""Debugging aid: phase timeline of the persistent decode engine (ITTS_ENGINE_STAMPS=1), one decode step.
Stamps per workgroup and block (100 MHz wall clock): 0 E1 gathered, 1 after B1, 2 q/k/v polled (attention WGs), 3 context
published, 4 E3 gathered, 5 E4 gathered, 6 E5 gathered, 7 after B5; compute side: 8 q/k/v published, 9 c_proj published,
10 c_fc published, 11 mlp.c_proj published."""
import os
import sys

import numpy as np
import torch

os.environ["ITTS_ENGINE_STAMPS"] = "1"
os.environ["ITTS_ENGINE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "index-tts-ipex_amd")):
    sys.path.insert(0, p)
from itts_hip import config as icfg, engine as ieng, synth  # noqa: E402

CFG = icfg.indextts_1_5()
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 240
eng = ieng.build_engine(CFG, "bf16", parts=("gpt",))
cond = eng.conditioning(torch.from_numpy(synth.prompt_mel(511, seed=7)))
text = np.stack([synth.text_ids(105, 11 + r, CFG.gpt.number_text_tokens) for r in range(rows)]).astype(np.int32)
eng.prefill(cond, text, 480, 10.0, True)
eng.decode(warm)
eng.debug(taps=True, no_graph=True)
eng.decode(1)
eng.fetch()
eng._exit()
st = eng.fetch_tap("eng_stamps").view(np.uint32).reshape(256, -1, 16).astype(np.int64)
eng.debug()
NL = st.shape[1]
t0 = st[:, 0, 0].min()
us = (st - t0) / 100.0
acu = np.array([(c < 240) and (c % 12 < rows) for c in range(256)])
print(f"rows {rows}, S ~ {139 + warm}; step span {us[:, :, 11].max():.1f} us, per block {(us[:, -1, 11].max() - us[:, 0, 0].min()) / NL:.2f} us")
names = ["E1 gathered", "after B1", "qkv polled (attn WG)", "ctx published (attn WG)", "E3 gathered", "E4 gathered", "E5 gathered",
         "after B5", "qkv published", "h1 published", "act published", "h2 published"]
for l in (1, NL // 2, NL - 2):
    base = us[:, l, 0].min()
    print(f"-- block {l} (t = 0 at the first WG's E1)")
    for i in (0, 1, 8, 2, 3, 4, 9, 5, 10, 6, 7, 11):
        v = us[:, l, i] - base
        if i in (2, 3):
            v = v[acu]
        print(f"   {names[i]:26s} min {v.min():6.2f}  median {np.median(v):6.2f}  max {v.max():6.2f}")
k0, k1 = (st[:, 1, 15] - t0) / 100.0, (st[:, 2, 15] - t0) / 100.0
print(f"-- launch: workgroups start {k0.min():.2f} .. {k0.max():.2f} us, first block's E1 gathered at 0 .. {us[:, 0, 0].max():.2f}; last h2 published "
      f"{us[:, -1, 11].max():.2f}, workgroups end {k1.min():.2f} .. {k1.max():.2f} us (head + sampler: {k1.max() - us[:, -1, 11].max():.2f} us; "
      f"whole launch {k1.max() - k0.min():.2f} us)")
# attention workgroups against the rest: mean over blocks 2..NL-2 of (stage time - block's first E1)
base = us[:, 2:NL - 1, 0].min(axis=0, keepdims=True)
rel = us[:, 2:NL - 1, :] - base[:, :, None]
print("-- attention workgroups vs the others (mean us since the block's first E1): " +
      ", ".join(f"{names[i]} {rel[acu][:, :, i].mean():.2f} / {rel[~acu][:, :, i].mean():.2f}" for i in (4, 9, 5, 10, 6, 11)))
clk = st[:, 0, 15]
print(f"shader clock over the launch: {np.median(clk) / 10.0:.0f} MHz (min {clk.min() / 10.0:.0f}, max {clk.max() / 10.0:.0f})")
d = us[:, 2:NL - 1, :]
print("-- c_fc phase of a compute wave (mean over WGs and blocks): E4 gathered (wave 0) -> after B4 %.2f -> LN2 done (2 barriers; mlp.c_proj DMA issued meanwhile) %.2f -> dots + wave sums %.2f -> gelu + publish %.2f us"
      % ((d[:, :, 12] - d[:, :, 5]).mean(), (d[:, :, 13] - d[:, :, 12]).mean(), (d[:, :, 14] - d[:, :, 13]).mean(), (d[:, :, 10] - d[:, :, 14]).mean()))
mx = us.max(axis=0)  # [NL][16] last WG
mx[:, 3] = us[acu][:, :, 3].max(axis=0)
seq = [(0, "E1 all gathered"), (8, "qkv all published"), (3, "ctx all published"), (4, "E3 all gathered"), (9, "h1 all published"),
       (5, "E4 all gathered"), (10, "act all published"), (6, "E5 all gathered"), (11, "h2 all published")]
prev = None
print("-- mean over blocks 2..NL-2 of (last WG at stage) - (last WG at previous stage)")
for i, nm in seq:
    if prev is not None:
        d = (mx[2:NL - 1, i] - mx[2:NL - 1, prev]).mean()
        print(f"   {nm:22s} +{d:5.2f} us")
    prev = i
print(f"   next block E1          +{(mx[3:NL, 0] - mx[2:NL - 1, 11]).mean():5.2f} us")
