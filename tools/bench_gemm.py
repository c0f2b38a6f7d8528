#!/usr/bin/env python3
"""A/B of the shift-GEMM kernels on the BigVGAN / GPT shapes of the bench (through the C ABI, HIP events):
gemm_glds (LDS-DMA staged, default where supported) vs gemm_mfma (register staged, ITTS_GEMM_FORCE_OLD=1).
    python tools/bench_gemm.py [--batch 1|32]   -> TFLOP/s per shape, both kernels"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "index-tts-ipex_amd"))
import torch  # noqa: E402

from itts_hip import lib as L  # noqa: E402


_WS = None


def run(lib, M, T, N, Cin, taps, dil, nphase=1, reps=10, old=False, p8=True, ksplit=None):
    """ksplit: None = itts_gemm (no workspace, never splits K); "auto" / int = itts_gemm_ws with a 64 MiB workspace (int forces
    that split count on eligible shapes through ITTS_GEMM_KSPLIT)."""
    global _WS
    dev = "cuda:0"
    A = torch.randn(M, Cin, device=dev).to(torch.bfloat16)
    W = (torch.randn(nphase, N, taps * Cin, device=dev) / (taps * Cin) ** 0.5).to(torch.bfloat16)
    Cout = torch.empty(M, nphase * N, device=dev, dtype=torch.bfloat16)
    g = L.GemmArgs()
    g.in_up, g.alpha = 1, 1.0
    g.A, g.W, g.C = A.data_ptr(), W.data_ptr(), Cout.data_ptr()
    g.M, g.N, g.Cin, g.taps, g.lda, g.ldc, g.T, g.dil, g.nphase = M, N, Cin, taps, Cin, nphase * N, T, dil, nphase
    g.pad_left = (taps - 1) * dil // 2 if nphase == 1 else 0
    g.dtype_a = g.dtype_w = g.dtype_c = L.BF16
    if old:
        os.environ["ITTS_GEMM_FORCE_OLD"] = "1"
    else:
        os.environ.pop("ITTS_GEMM_FORCE_OLD", None)
    os.environ["ITTS_GEMM_P8"] = "1" if p8 else "0"
    which = int(lib.itts_gemm_which(C.byref(g)))
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    if ksplit is None:
        call = lambda: lib.itts_gemm(C.byref(g), s)  # noqa: E731
    else:
        if _WS is None:
            _WS = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
        if ksplit == "auto":
            os.environ.pop("ITTS_GEMM_KSPLIT", None)
        else:
            os.environ["ITTS_GEMM_KSPLIT"] = str(ksplit)
        ns = int(lib.itts_gemm_ksplit(C.byref(g), _WS.numel()))
        if ns > 1:
            which = 30 + ns
        call = lambda: lib.itts_gemm_ws(C.byref(g), _WS.data_ptr(), _WS.numel(), s)  # noqa: E731
    for _ in range(2):
        L.check(call())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    os.environ.pop("ITTS_GEMM_KSPLIT", None)
    us = e0.elapsed_time(e1) * 1e3 / reps
    return us, 2.0 * M * N * nphase * taps * Cin / us / 1e6, which


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--ksplit", action="store_true", help="K-split sweep (itts_gemm_ws): no split / auto / 2 / 4 / 8 per shape")
    a = ap.parse_args()
    lib = L.load()
    rows = 2 * a.batch  # sentences
    T0 = 480
    shapes = [("conv_pre 1280->1536 k7", rows * T0, T0, 1536, 1280, 7, 1, 1)]
    ch, Tc = 1536, T0
    for i, (u, k) in enumerate(zip([4, 4, 4, 4], [8, 8, 4, 4])):
        shapes.append((f"up{i} {ch}->{ch // 2} x{u}", rows * Tc, Tc, ch // 2, ch, k // u, -1, u))
        ch //= 2
        Tc *= u
        if ch >= 192:
            for kk, d in ((3, 1), (7, 3), (11, 5)):
                shapes.append((f"amp{i} C={ch} k{kk} d{d}", rows * Tc, Tc, ch, ch, kk, d, 1))
    n = 32 + 107 + 482
    for nm, N, K in (("latent c_attn", 3840, 1280), ("latent c_proj", 1280, 1280), ("latent c_fc", 5120, 1280), ("latent proj2", 1280, 5120)):
        shapes.append((nm, rows * n, rows * n, N, K, 1, 1, 1))
    names = {0: "valu", 1: "mfma", 2: "glds", 3: "p8", 4: "convlds"}
    names.update({30 + k: "p8/%d" % k for k in range(2, 9)})
    if a.ksplit:
        extra = [("prefill proj2 (2 x 142 rows)", 284, 284, 1280, 5120, 1, 1, 1), ("prefill c_fc", 284, 284, 5120, 1280, 1, 1, 1)]
        print(f"{'shape':30s} {'M':>7s}  " + "  ".join(f"{h:>14s}" for h in ("no split", "auto", "S=2", "S=4", "S=8")))
        for nm, M, T, N, Cin, taps, dil, nph in shapes + extra:
            cols = []
            for ks in (None, "auto", 2, 4, 8):
                us, tf, wh = run(lib, M, T, N, Cin, taps, dil, nph, ksplit=ks)
                cols.append(f"{names[wh]:>5s} {us:7.1f}us")
            print(f"{nm:30s} {M:7d}  " + "  ".join(cols), flush=True)
        return
    print(f"{'shape':28s} {'M':>9s}  {'default':>8s} {'us':>9s} {'TF/s':>7s}   {'no-p8':>7s} {'us':>9s} {'TF/s':>7s}   {'old us':>9s} {'TF/s':>7s}  default vs no-p8")
    for nm, M, T, N, Cin, taps, dil, nph in shapes:
        new = run(lib, M, T, N, Cin, taps, dil, nph)
        mid = run(lib, M, T, N, Cin, taps, dil, nph, p8=False)
        old = run(lib, M, T, N, Cin, taps, dil, nph, old=True)
        print(f"{nm:28s} {M:9d}  {names[new[2]]:>8s} {new[0]:9.1f} {new[1]:7.1f}   {names[mid[2]]:>7s} {mid[0]:9.1f} {mid[1]:7.1f}   {old[0]:9.1f} {old[1]:7.1f}  "
              f"{mid[0] / new[0]:5.2f}x", flush=True)


if __name__ == "__main__":
    main()
