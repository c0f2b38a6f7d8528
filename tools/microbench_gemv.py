"""GPU micro-benchmark of the decode GEMV: achieved GB/s per shape (hipGraph of 20 back-to-back launches over
distinct weight slabs so nothing is cache-resident)."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "index-tts-ipex_amd"))
from itts_hip import lib as L  # noqa: E402

lib = L.load()
dev = "cuda:0"
st = torch.cuda.Stream()
REP = 20
for (N, K, B, pro) in [(1280, 1280, 2, 0), (3840, 1280, 2, 1), (5120, 1280, 2, 1), (1280, 5120, 2, 0), (8194, 1280, 2, 2),
                       (40960, 1280, 2, 1), (20480, 5120, 2, 0), (3840, 1280, 1, 1), (5120, 1280, 4, 1)]:
    W = torch.randn(REP, N, K, device=dev).to(torch.bfloat16)
    X = torch.randn(B, K, device=dev)
    Y = torch.zeros(B, N, device=dev)
    g = torch.ones(K, device=dev)
    b = torch.zeros(K, device=dev)
    bias = torch.zeros(N, device=dev)
    for ver in (2, 1):
        if ver == 1 and pro == 2:
            continue

        def launch():
            for r in range(REP):
                L.check(lib.itts_gemv(Y.data_ptr(), X.data_ptr(), W[r].data_ptr(), bias.data_ptr(), B, N, K, 0, 0, pro,
                                      g.data_ptr(), b.data_ptr(), g.data_ptr(), b.data_ptr(), L.BF16, ver,
                                      C.c_void_p(st.cuda_stream)))

        with torch.cuda.stream(st):
            launch()
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=st):
                launch()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            gr.replay()
            torch.cuda.synchronize()
            e0.record(st)
            for _ in range(10):
                gr.replay()
            e1.record(st)
            torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (10 * REP)
        print(f"v{ver} N={N:6d} K={K:5d} B={B} prologue={pro}: {us:7.2f} us/launch  {N * K * 2 / us / 1e3:8.1f} GB/s")
