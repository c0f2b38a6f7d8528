#!/usr/bin/env python3
"""Does the 64-row decode step gain from running as two 32-row halves on two HIP streams?  (The launch path's step is a chain
of ~170 dependent launches of 5-25 us; two independent chains can fill each other's ramps and drains.)
Times K decode steps of: one engine at 64 rows; one engine at 32 rows; two engines at 32 rows each, concurrently."""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "index-tts-ipex_amd"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from itts_hip import config as icfg, engine as E, synth  # noqa: E402


def main():
    cfg = icfg.indextts_1_5()
    g = cfg["gpt"]
    K, T, L = 200, 488, 107
    engs = [E.build_engine(cfg, "bf16", parts=("gpt",), max_batch=64) for _ in range(2)]
    D = g["model_dim"]
    cond = torch.randn(32, D, device="cuda:0")

    def texts(rows, seed):
        return np.stack([synth.text_ids(L, seed + k, g["number_text_tokens"]) for k in range(rows)]).astype(np.int32)

    def run(pairs):
        for e, rows in pairs:
            e.prefill(cond, texts(rows, 11), T, 10.0, True)
            e.decode(8)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=lambda e=e: e.decode(K)) for e, _ in pairs]
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K * 1e3
        for e, _ in pairs:
            e.fetch()
            e._exit()
        return dt

    for name, pairs in (("1 x 64 rows", [(engs[0], 64)]), ("1 x 32 rows", [(engs[0], 32)]), ("2 x 32 rows, two streams", [(engs[0], 32), (engs[1], 32)]),
                        ("2 x 16 rows", [(engs[0], 16), (engs[1], 16)]), ("1 x 16 rows", [(engs[0], 16)])):
        ms = [run(pairs) for _ in range(2)]
        print(f"{name:28s} {min(ms):.3f} ms / step", flush=True)


if __name__ == "__main__":
    main()
