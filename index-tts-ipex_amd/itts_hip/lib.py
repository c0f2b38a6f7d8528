"""ctypes binding of libitts_hip.so (C ABI declared in include/itts_hip.h).

The product path has NO CPU fallback: if the shared library is missing or a call fails, this module
raises RuntimeError (the reference's loader does the same when its CUDA extension cannot be built,
indextts/BigVGAN/alias_free_activation/cuda/load.py:51-52,82-87).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.normpath(os.path.join(_HERE, "..", "csrc"))
LIB_PATH = os.environ.get("ITTS_HIP_LIB", os.path.join(CSRC, "libitts_hip.so"))
LIB_PATH_F16 = os.environ.get("ITTS_HIP_LIB_F16", os.path.join(CSRC, "libitts_hip_f16.so"))  # the same sources, IEEE half storage

F32, BF16 = 0, 1
FP8 = 4  # OCP e4m3fn bytes (GPT decode weights, BASELINE config 5)
F16 = 5  # IEEE half, operator-level boundary only (itts_snake_aa_fwd)
ACT_NONE, ACT_RELU, ACT_SILU, ACT_GELU_NEW, ACT_GELU_ERF, ACT_TANH, ACT_SIGMOID = range(7)  # csrc/itts_common.h enum Act
ACT = {"none": 0, "relu": 1, "silu": 2, "gelu_new": 3, "gelu_erf": 4, "tanh": 5, "sigmoid": 6}

vp, i32, f32, i64 = C.c_void_p, C.c_int, C.c_float, C.c_int64


class GemmArgs(C.Structure):
    _fields_ = [("A", vp), ("W", vp), ("C", vp),
                ("M", i32), ("N", i32), ("Cin", i32), ("taps", i32), ("lda", i32), ("ldc", i32), ("T", i32),
                ("dil", i32), ("pad_left", i32), ("pad_mode", i32), ("in_up", i32), ("nphase", i32),
                ("phase_shift", i32 * 8),
                ("bias", vp), ("bias_bstride", i32), ("act", i32),
                ("scale", vp), ("shift", vp), ("act2", i32),
                ("R", vp), ("ldr", i32), ("alpha", f32),
                ("ADD", vp), ("ldadd", i32), ("beta", f32),
                ("dtype_a", i32), ("dtype_w", i32), ("dtype_c", i32), ("force_simple", i32)]


class Config(C.Structure):
    _fields_ = [("dtype", i32),
                ("model_dim", i32), ("layers", i32), ("heads", i32), ("max_mel_tokens", i32), ("max_text_tokens", i32),
                ("number_text_tokens", i32), ("number_mel_codes", i32),
                ("start_mel_token", i32), ("stop_mel_token", i32), ("start_text_token", i32), ("stop_text_token", i32),
                ("cond_latents", i32),
                ("cond_dim", i32), ("cond_ff", i32), ("cond_heads", i32), ("cond_blocks", i32), ("cond_idim", i32),
                ("perc_inner", i32), ("perc_layers", i32),
                ("bv_gpt_dim", i32), ("bv_init_ch", i32), ("bv_num_up", i32), ("bv_up_rates", i32 * 8),
                ("bv_up_kernels", i32 * 8),
                ("bv_num_res", i32), ("bv_res_kernels", i32 * 4), ("bv_res_dils", (i32 * 4) * 4), ("bv_num_dil", i32),
                ("bv_spk_dim", i32), ("bv_num_mels", i32),
                ("ec_channels", i32 * 5), ("ec_kernels", i32 * 5), ("ec_dils", i32 * 5), ("ec_att", i32),
                ("ec_scale", i32), ("ec_se", i32),
                ("dv_channels", i32), ("dv_tokens", i32), ("dv_hidden", i32), ("dv_resblocks", i32),
                ("dv_codebook", i32), ("dv_layers", i32), ("dv_kernel", i32),
                ("max_batch", i32)]


_lib = None
_lib_f16 = None

_PROTOS = {
    "itts_last_error": (C.c_char_p, []),
    "itts_abi_version": (i32, []),
    "itts_half_is_f16": (i32, []),
    "itts_gpt_set_beam_returns": (i32, [vp, i32]),
    "itts_snake_aa_fwd": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "itts_gemm": (i32, [C.POINTER(GemmArgs), vp]),
    "itts_gemm_which": (i32, [C.POINTER(GemmArgs)]),
    "itts_gemm_ws": (i32, [C.POINTER(GemmArgs), vp, C.c_size_t, vp]),
    "itts_gemm_ksplit": (i32, [C.POINTER(GemmArgs), C.c_size_t]),
    "itts_layernorm": (i32, [vp, i32, vp, i32, vp, vp, i32, i32, f32, vp]),
    "itts_attention": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, f32, i32, vp, i32, vp]),
    "itts_gemv": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, i32, i32, vp]),
    "itts_skinny_gemm": (i32, [vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, i32, vp]),
    "itts_retile_weights": (i32, [vp, vp, i32, i32, vp]),
    "itts_ln_rows_bf16": (i32, [vp, vp, vp, vp, i32, i32, f32, i32, vp, i32, vp, i32, vp]),
    "itts_transpose": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "itts_engine_create": (i32, [C.POINTER(Config), C.POINTER(vp)]),
    "itts_engine_destroy": (None, [vp]),
    "itts_engine_bind_tensor": (i32, [vp, C.c_char_p, vp, i32, i32, C.POINTER(i64)]),
    "itts_engine_finalize": (i32, [vp]),
    "itts_conditioning": (i32, [vp, vp, i32, vp, vp]),
    "itts_conditioning_padded": (i32, [vp, vp, i32, i32, vp, vp]),
    "itts_ecapa": (i32, [vp, vp, i32, i32, vp, vp]),
    "itts_gpt_prefill": (i32, [vp, vp, vp, i32, i32, i32, f32, i32, vp]),
    "itts_gpt_set_sampling": (i32, [vp, i32, i32, f32, f32, vp, i64]),
    "itts_gpt_set_forced": (i32, [vp, vp, i32, i32]),
    "itts_gpt_set_input_tokens": (i32, [vp, vp, i32, i32]),
    "itts_gpt_decode_mode": (i32, [vp]),
    "itts_gpt_set_host_sampling": (i32, [vp, i32]),
    "itts_gpt_set_cond_per_row": (i32, [vp, i32]),
    "itts_gpt_commit": (i32, [vp, vp, vp]),
    "itts_gpt_beam_state": (i32, [vp, vp, vp, vp, C.POINTER(i32), vp]),
    "itts_gpt_commit_beams": (i32, [vp, vp, vp, vp, vp]),
    "itts_gpt_set_typical": (i32, [vp, f32]),
    "itts_gpt_set_beams": (i32, [vp, i32, i32, i32, f32, f32, f32, vp, i64]),
    "itts_gpt_set_beam_sample": (i32, [vp, i32, i32, f32, f32, vp, i64]),
    "itts_gpt_decode": (i32, [vp, i32, vp]),
    "itts_gpt_status": (i32, [vp, C.POINTER(i32), C.POINTER(i32), vp]),
    "itts_gpt_fetch": (i32, [vp, vp, vp, vp]),
    "itts_gpt_latent": (i32, [vp, vp, vp, i32, vp, i32, vp, vp]),
    "itts_gpt_latent_batch": (i32, [vp, vp, vp, vp, vp, vp, i32, vp, vp]),
    "itts_bigvgan": (i32, [vp, vp, vp, i32, i32, vp, vp]),
    "itts_dvae_decode": (i32, [vp, vp, i32, i32, vp, vp]),
    "itts_dvae_encode": (i32, [vp, vp, i32, i32, vp, vp]),
    "itts_debug_fetch": (i64, [vp, C.c_char_p, vp, i64]),
    "itts_debug_enable": (i32, [vp, i32]),
}


def exported_symbols():
    """Every entry point include/itts_hip.h declares (used by the CPU load/export test)."""
    return sorted(_PROTOS)


def load(half: str = "bf16"):
    """dlopen the library and attach prototypes.  No GPU work happens here.  half = "f16": the build whose 16-bit storage type
    is IEEE binary16 (libitts_hip_f16.so) - a second, independent library object with its own engines and error state."""
    global _lib, _lib_f16
    if half == "f16":
        if _lib_f16 is None:
            _lib_f16 = _open(LIB_PATH_F16)
            assert _lib_f16.itts_half_is_f16() == 1
        return _lib_f16
    if _lib is not None:
        return _lib
    _lib = _open(LIB_PATH)
    assert _lib.itts_half_is_f16() == 0
    return _lib


def _open(path):
    if not os.path.exists(path):
        raise RuntimeError(
            f"{os.path.basename(path)} not found at {path}: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C {CSRC}` (there is no CPU fallback)")
    # torch ships its own libamdhip64; the process must run ONE HIP runtime, and device pointers / streams come from
    # torch, so torch's copy has to be the one already loaded when this library resolves its HIP symbols (loading
    # libitts_hip first pulls in /opt/rocm's runtime and the engine then sees no device)
    import torch  # noqa: F401

    lib = C.CDLL(path)
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


E_HANDOFF = -6  # include/itts_hip.h ITTS_E_HANDOFF


class HandoffTimeout(RuntimeError):
    """A hand-off wait inside the persistent decode engine gave up: this generation's codes are not valid, the engine
    object has switched to the launch path; generating again is safe (Engine.generate does it once)."""


def check(status: int, what: str = "", lib=None):
    if status != 0:
        msg = (lib or load()).itts_last_error().decode("utf-8", "replace")
        if status == E_HANDOFF:
            raise HandoffTimeout(f"libitts_hip {what} failed ({status}): {msg}")
        raise RuntimeError(f"libitts_hip {what} failed ({status}): {msg}")
