"""Vocoder (and latent-pass) only, at BASELINE config 3's shapes: 64 sentences x 480 frames through BigVGAN in one batched launch
sequence - the workload of tools/prof_vocoder.sh (rocprofv3 kernel summary of the vocoder kernels without the 479 decode steps)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "index-tts-ipex_amd")):
    sys.path.insert(0, p)
from itts_hip import config as icfg, engine as ieng, synth  # noqa: E402

CFG = icfg.indextts_1_5()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 480
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
eng = ieng.build_engine(CFG, "bf16", parts=("bigvgan",), max_batch=128)
mel = torch.from_numpy(synth.prompt_mel(511, seed=7)).cuda()
spk = eng.ecapa(mel.transpose(1, 2)).expand(B, -1).contiguous()
lat = (torch.randn(B, T, CFG.bigvgan.gpt_dim, device="cuda") * 1.0).to(torch.bfloat16)
for _ in range(reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    wav = eng.bigvgan(lat, spk)
    torch.cuda.synchronize()
    print(f"bigvgan {B} x {T} frames: {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
