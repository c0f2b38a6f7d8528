"""Loader of the fused anti-aliased activation op.

The reference JIT-compiles a CUDA extension here (`load()` -> module `anti_alias_activation_cuda` with
`forward(input, up_filter, down_filter, alpha, beta) -> Tensor`,
/root/reference/indextts/BigVGAN/alias_free_activation/cuda/load.py:49-133, anti_alias_activation.cpp:19-23).
This drop-in returns an object with the same `forward`, bound to `itts_snake_aa_fwd` of libitts_hip
(built in-tree by `make -C index-tts-ipex_amd/csrc`).  Same contract: contiguous [B, C, T] input in
fp32 / bf16, fp32 filters [12], LOG-scale fp32 alpha / beta [C]; the output is allocated here like
`torch::empty_like(input)` in fwd_cuda (.cu:222-225).  Raises RuntimeError when the library or a GPU is
missing (the reference raises when nvcc / CUDA are missing, load.py:51-52,82-87)."""
from __future__ import annotations

import ctypes as C

import torch

from itts_hip import lib as L


class _AntiAliasActivation:
    def __init__(self):
        self._lib = L.load()

    def forward(self, inputs: torch.Tensor, up_ftr: torch.Tensor, down_ftr: torch.Tensor, alpha: torch.Tensor,
                beta: torch.Tensor) -> torch.Tensor:
        if not inputs.is_cuda:
            raise RuntimeError("anti_alias_activation_cuda.forward: input must live on the GPU")
        if inputs.dim() != 3 or not inputs.is_contiguous():
            raise RuntimeError("anti_alias_activation_cuda.forward: expected a contiguous [B, C, T] tensor")
        if inputs.dtype == torch.float32:
            dt = L.F32
        elif inputs.dtype == torch.bfloat16:
            dt = L.BF16
        elif inputs.dtype == torch.float16:  # the reference's GPU default is .half() (infer.py:44,52)
            dt = L.F16
        else:  # the reference dispatches float/half/bfloat16 and AT_ERRORs on anything else (type_shim.h:20-43)
            raise RuntimeError(f"anti_alias_activation_cuda.forward: unsupported dtype {inputs.dtype}")
        B, Cc, T = inputs.shape
        dev = inputs.device
        up = up_ftr.reshape(-1).to(device=dev, dtype=torch.float32).contiguous()
        dn = down_ftr.reshape(-1).to(device=dev, dtype=torch.float32).contiguous()
        al = alpha.reshape(-1).to(device=dev, dtype=torch.float32).contiguous()
        be = beta.reshape(-1).to(device=dev, dtype=torch.float32).contiguous()
        if up.numel() != 12 or dn.numel() != 12 or al.numel() != Cc or be.numel() != Cc:
            raise RuntimeError("anti_alias_activation_cuda.forward: filters must hold 12 taps, alpha/beta C values")
        out = torch.empty_like(inputs)
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        L.check(self._lib.itts_snake_aa_fwd(out.data_ptr(), inputs.data_ptr(), up.data_ptr(), dn.data_ptr(), al.data_ptr(),
                                            be.data_ptr(), B, Cc, T, dt, 0, stream), "itts_snake_aa_fwd")
        for t in (up, dn, al, be):
            t.record_stream(torch.cuda.current_stream(dev))
        return out


def load():
    if not torch.cuda.is_available():
        raise RuntimeError("anti_alias_activation: no GPU available")
    return _AntiAliasActivation()
