"""Conditioning encoder, speaker encoder, and the pair with the speaker encoder on the engine's side stream (Engine.ecapa(overlap=True)):
ms per call at IndexTTS-1.5 sizes, prompt of 511 frames.   python tools/t_ecapa.py"""
import sys, time, torch, numpy as np
sys.path.insert(0, "/root/repo/index-tts-ipex_amd")
from itts_hip import config as icfg, engine as ieng, synth
CFG = icfg.indextts_1_5()
eng = ieng.build_engine(CFG, "bf16", parts=("gpt", "bigvgan"))
mel = torch.from_numpy(synth.prompt_mel(511, seed=7)).cuda()
for name, fn in (("conditioning", lambda: eng.conditioning(mel)), ("ecapa", lambda: eng.ecapa(mel.transpose(1, 2)))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        fn()
    eng._exit()
    torch.cuda.synchronize()
    print(name, (time.perf_counter() - t0) / 10 * 1e3, "ms")


def both():
    spk = eng.ecapa(mel.transpose(1, 2), overlap=True)
    cond = eng.conditioning(mel)
    eng._join_side()
    return spk, cond


for _ in range(3):
    both()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    both()
eng._exit()
torch.cuda.synchronize()
print("ecapa on the side stream + conditioning", (time.perf_counter() - t0) / 10 * 1e3, "ms")
