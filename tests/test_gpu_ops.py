"""GPU: operator-level parity of the HIP kernels (through the C ABI) against the oracle / torch-fp32 CPU."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from itts_hip import lib as L  # noqa: E402
from itts_hip import pack, prng  # noqa: E402
from oracle import vocoder as ovoc  # noqa: E402

DEV = "cuda:0"


def rnd(name, shape, std=1.0):
    return torch.from_numpy(prng.tensor(name, 5, shape, std=std))


def relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.fixture(scope="module")
def lib():
    return L.load()


@pytest.mark.parametrize("tag", ["a", "b", "c"])
@pytest.mark.parametrize("layout", [0, 1])
def test_snake_golden(lib, gold, tag, layout):
    g = gold(f"micro_act1d_{tag}")
    x = torch.from_numpy(g["x"]).to(DEV)
    B, Cc, T = x.shape
    xin = x.contiguous() if layout == 0 else x.transpose(1, 2).contiguous()
    y = torch.empty_like(xin)
    filt = torch.from_numpy(g["filt"]).to(DEV)
    al, be = torch.from_numpy(g["alpha"]).to(DEV), torch.from_numpy(g["beta"]).to(DEV)
    L.check(lib.itts_snake_aa_fwd(y.data_ptr(), xin.data_ptr(), filt.data_ptr(), filt.data_ptr(), al.data_ptr(),
                                  be.data_ptr(), B, Cc, T, L.F32, layout, stream()))
    out = y if layout == 0 else y.transpose(1, 2)
    assert relerr(out, torch.from_numpy(g["y"])) < 2e-5


@pytest.mark.parametrize("shape", [(2, 768, 100), (1, 24, 4099), (3, 5, 33), (2, 48, 1000), (1, 96, 700), (2, 192, 333),
                                   (1, 1536, 50), (2, 24, 7), (1, 40, 321)])
@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_snake_vs_oracle(lib, shape, dt):
    B, Cc, T = shape
    x = rnd(f"snk{shape}", shape, 1.5)
    al, be = rnd("snk.a", (Cc,), 0.4), rnd("snk.b", (Cc,), 0.4)
    tdt = torch.float32 if dt == "f32" else torch.bfloat16
    xq = x.to(tdt)
    ref = ovoc.activation1d(xq.float(), al, be)
    filt = ovoc.kaiser_sinc_filter12().to(DEV)
    ald, bed = al.to(DEV), be.to(DEV)  # keep alive: data_ptr() of a temporary dangles
    for layout in (0, 1):
        xin = (xq if layout == 0 else xq.transpose(1, 2)).contiguous().to(DEV)
        y = torch.empty_like(xin)
        L.check(lib.itts_snake_aa_fwd(y.data_ptr(), xin.data_ptr(), filt.data_ptr(), filt.data_ptr(),
                                      ald.data_ptr(), bed.data_ptr(), B, Cc, T,
                                      L.F32 if dt == "f32" else L.BF16, layout, stream()))
        out = (y if layout == 0 else y.transpose(1, 2)).float()
        assert relerr(out, ref) < (2e-5 if dt == "f32" else 1e-2)


def run_gemm(lib, A, W, Cshape, dt_a, dt_w, dt_c, force_simple=0, **kw):
    kw_ws = {"ws": kw.pop("ws", None)}
    g = L.GemmArgs()
    for k in ("dil", "in_up", "nphase", "taps"):
        setattr(g, k, 1)
    g.alpha = 1.0
    g.A, g.W = A.data_ptr(), W.data_ptr()
    Cout = torch.empty(Cshape, dtype=torch.float32 if dt_c == L.F32 else torch.bfloat16, device=DEV)
    g.C = Cout.data_ptr()
    keep = []
    for k, v in kw.items():
        if isinstance(v, torch.Tensor):
            keep.append(v)
            v = v.data_ptr()
        if k == "phase_shift":
            for i, s in enumerate(v):
                g.phase_shift[i] = s
            continue
        setattr(g, k, v)
    g.dtype_a, g.dtype_w, g.dtype_c, g.force_simple = dt_a, dt_w, dt_c, force_simple
    run_gemm.which = int(lib.itts_gemm_which(C.byref(g)))  # kernel family the dispatcher took (include/itts_hip.h)
    ws = kw_ws.get("ws")
    if ws is not None:  # caller-owned K-split workspace (itts_gemm_ws)
        run_gemm.ksplit = int(lib.itts_gemm_ksplit(C.byref(g), ws.numel() * ws.element_size()))
        L.check(lib.itts_gemm_ws(C.byref(g), ws.data_ptr(), ws.numel() * ws.element_size(), stream()))
    else:
        L.check(lib.itts_gemm(C.byref(g), stream()))
    torch.cuda.synchronize()
    return Cout


CONV_CASES = [
    # B, T, Cin, Cout, k, dil, mode
    (2, 37, 32, 48, 3, 1, "zeros"), (1, 50, 64, 64, 7, 3, "zeros"), (2, 41, 24, 24, 11, 5, "zeros"),
    (1, 33, 100, 64, 5, 1, "reflect"), (2, 29, 64, 64, 3, 4, "reflect"), (1, 64, 128, 1, 7, 1, "zeros"),
    (1, 130, 256, 192, 3, 1, "zeros"),
]


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("simple", [1, 0])
def test_conv1d(lib, case, dt, simple):
    B, T, Cin, Cout, k, dil, mode = case
    tdt = torch.float32 if dt == "f32" else torch.bfloat16
    ldt = L.F32 if dt == "f32" else L.BF16
    x = rnd(f"cv.x{case}", (B, Cin, T)).to(tdt)
    w = rnd(f"cv.w{case}", (Cout, Cin, k), 1.0 / np.sqrt(Cin * k)).to(tdt)
    bias = rnd(f"cv.b{case}", (Cout,), 0.1)
    pad = dil * (k - 1) // 2
    xp = F.pad(x.float(), (pad, pad), mode="reflect") if mode == "reflect" else F.pad(x.float(), (pad, pad))
    ref = F.conv1d(xp, w.float(), bias, dilation=dil)
    res = rnd(f"cv.r{case}", (B, Cout, T)).to(tdt)
    ref = F.relu(ref) + res.float()
    A = x.transpose(1, 2).contiguous().to(DEV)
    W = torch.from_numpy(pack.conv_w(w.float().numpy())).to(tdt).to(DEV)
    R = res.transpose(1, 2).contiguous().to(DEV)
    out = run_gemm(lib, A, W, (B, T, Cout), ldt, ldt, ldt, simple, M=B * T, N=Cout, Cin=Cin, taps=k, lda=Cin, ldc=Cout,
                   T=T, dil=dil, pad_left=pad, pad_mode=1 if mode == "reflect" else 0, bias=bias.to(DEV), act=1, R=R,
                   ldr=Cout)
    assert relerr(out.float().transpose(1, 2), ref) < (2e-5 if dt == "f32" else 2e-2)


@pytest.mark.parametrize("case", [(2, 19, 64, 32, 8, 4), (1, 23, 32, 16, 4, 4), (2, 17, 48, 24, 4, 2), (1, 9, 128, 64, 8, 4)])
@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("simple", [1, 0])
def test_conv_transpose1d(lib, case, dt, simple):
    B, T, Cin, Cout, k, u = case
    p = (k - u) // 2
    tdt = torch.float32 if dt == "f32" else torch.bfloat16
    ldt = L.F32 if dt == "f32" else L.BF16
    x = rnd(f"ct.x{case}", (B, Cin, T)).to(tdt)
    w = rnd(f"ct.w{case}", (Cin, Cout, k), 1.0 / np.sqrt(Cin * k / u)).to(tdt)
    bias = rnd(f"ct.b{case}", (B, Cout), 0.1)  # per-batch bias (speaker conditioning folded in)
    ref = F.conv_transpose1d(x.float(), w.float(), None, stride=u, padding=p) + bias[:, :, None]
    A = x.transpose(1, 2).contiguous().to(DEV)
    W = torch.from_numpy(pack.convT_w(w.float().numpy(), u, p)).to(tdt).to(DEV)
    out = run_gemm(lib, A, W, (B, T * u, Cout), ldt, ldt, ldt, simple, M=B * T, N=Cout, Cin=Cin, taps=k // u, lda=Cin,
                   ldc=u * Cout, T=T, dil=-1, pad_left=0, nphase=u, phase_shift=[(ph + p) // u for ph in range(u)],
                   bias=bias.to(DEV), bias_bstride=Cout)
    assert relerr(out.float().transpose(1, 2), ref) < (2e-5 if dt == "f32" else 2e-2)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_upsampled_conv_and_linear(lib, dt):
    tdt = torch.float32 if dt == "f32" else torch.bfloat16
    ldt = L.F32 if dt == "f32" else L.BF16
    B, T, Cin, Cout = 2, 11, 64, 32
    x = rnd("uc.x", (B, Cin, T)).to(tdt)
    w = rnd("uc.w", (Cout, Cin, 3), 0.1).to(tdt)
    ref = F.conv1d(F.interpolate(x.float(), scale_factor=2, mode="nearest"), w.float(), None, padding=1)
    A = x.transpose(1, 2).contiguous().to(DEV)
    W = torch.from_numpy(pack.conv_w(w.float().numpy())).to(tdt).to(DEV)
    out = run_gemm(lib, A, W, (B, 2 * T, Cout), ldt, ldt, ldt, 0, M=B * 2 * T, N=Cout, Cin=Cin, taps=3, lda=Cin, ldc=Cout,
                   T=2 * T, pad_left=1, in_up=2)
    assert relerr(out.float().transpose(1, 2), ref) < (2e-5 if dt == "f32" else 2e-2)
    # linear with odd K and alpha/affine epilogue, fp32 output
    M, K, N = 37, 3413, 130
    a = rnd("ln.a", (M, K)).to(tdt)
    wl = rnd("ln.w", (N, K), 0.02).to(tdt)
    sc, sh = rnd("ln.sc", (N,), 0.2) + 1, rnd("ln.sh", (N,), 0.2)
    ref = torch.tanh(F.silu(a.float() @ wl.float().T) * sc + sh) * 0.5
    out = run_gemm(lib, a.to(DEV), wl.to(DEV), (M, N), ldt, ldt, L.F32, 0, M=M, N=N, Cin=K, lda=K, ldc=N, act=2,
                   scale=sc.to(DEV), shift=sh.to(DEV), act2=5, alpha=0.5)
    assert relerr(out, ref) < (5e-5 if dt == "f32" else 2e-2)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_attention(lib, dt):
    tdt = torch.float32 if dt == "f32" else torch.bfloat16
    ldt = L.F32 if dt == "f32" else L.BF16
    B, H, S, dh = 2, 3, 150, 64
    D = H * dh
    qkv = rnd("at.qkv", (B, S, 3 * D)).to(tdt)
    kvs = torch.tensor([0, 7], dtype=torch.int32)
    q, k, v = [t.float().view(B, S, H, dh).transpose(1, 2) for t in qkv.split(D, dim=2)]
    att = (q @ k.transpose(-1, -2)) / 8.0
    mask = torch.tril(torch.ones(S, S, dtype=torch.bool))[None, None].expand(B, 1, S, S).clone()
    mask[1, :, :, :7] = False
    att = att.masked_fill(~mask, float("-inf"))
    ref = (torch.softmax(att, -1).nan_to_num(0.0) @ v).transpose(1, 2).reshape(B, S, D)
    x = qkv.to(DEV)
    kvd = kvs.to(DEV)
    o = torch.empty(B, S, D, dtype=tdt, device=DEV)
    es = x.element_size()
    L.check(lib.itts_attention(o.data_ptr(), x.data_ptr(), x.data_ptr() + D * es, x.data_ptr() + 2 * D * es, B, H, S, S,
                               dh, dh, 3 * D, 3 * D, 3 * D, D, 0.125, 1, kvd.data_ptr(), ldt, stream()))
    torch.cuda.synchronize()
    # rows that are themselves padding (query < kv_start) are don't-care
    assert relerr(o.float()[0], ref[0]) < (2e-5 if dt == "f32" else 2e-2)
    assert relerr(o.float()[1, 7:], ref[1, 7:]) < (2e-5 if dt == "f32" else 2e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(1, 20, 621, 621, 1, 0), (2, 4, 96, 200, 1, 5), (1, 16, 32, 290, 0, 0), (3, 2, 17, 17, 1, 3),
                                  (2, 5, 130, 130, 0, 0)])
def test_attention_mfma_vs_simple(lib, case):
    """bf16 flash attention on MFMA vs fp64 torch math: Sq != Sk (causal shifted), left padding, ragged tiles."""
    B, H, Sq, Sk, causal, pad = case
    dh, D = 64, H * 64
    q = rnd("am.q", (B, Sq, D)).to(torch.bfloat16)
    kv = rnd("am.kv", (B, Sk, 2 * D)).to(torch.bfloat16)
    kvs = torch.tensor([(pad if b == B - 1 else 0) for b in range(B)], dtype=torch.int32)
    qf = q.double().view(B, Sq, H, dh).transpose(1, 2)
    kf, vf = [t.double().view(B, Sk, H, dh).transpose(1, 2) for t in kv.split(D, dim=2)]
    att = (qf @ kf.transpose(-1, -2)) * 0.125
    j = torch.arange(Sk)[None, :]
    i = torch.arange(Sq)[:, None]
    mask = torch.ones(B, 1, Sq, Sk, dtype=torch.bool)
    if causal:
        mask &= (j <= i + (Sk - Sq))[None, None]
    for b in range(B):
        mask[b] &= (j >= int(kvs[b]))[None]
    ref = (torch.softmax(att.masked_fill(~mask, float("-inf")), -1).nan_to_num(0.0) @ vf).transpose(1, 2).reshape(B, Sq, D)
    qd, kvd, kd = q.to(DEV), kv.to(DEV), kvs.to(DEV)
    o = torch.full((B, Sq, D), float("nan"), dtype=torch.bfloat16, device=DEV)
    L.check(lib.itts_attention(o.data_ptr(), qd.data_ptr(), kvd.data_ptr(), kvd.data_ptr() + D * 2, B, H, Sq, Sk, dh, dh,
                               D, 2 * D, 2 * D, D, 0.125, causal, kd.data_ptr(), L.BF16, stream()))
    torch.cuda.synchronize()
    valid = mask.any(-1)[:, 0]  # [B, Sq] rows with at least one visible key
    got = o.float().cpu()
    assert torch.isfinite(got).all()
    for b in range(B):
        assert relerr(got[b][valid[b]], ref[b][valid[b]].float()) < 1.5e-2


def to_tiles(x, BT):
    """row-major [B, K] -> MFMA-fragment tiles (csrc/itts_decode.h tile_off), padded rows poisoned with NaN"""
    B, K = x.shape
    xp = torch.full((BT * 16, K), float("nan"), dtype=x.dtype)
    xp[:B] = x
    return xp.view(BT, 16, K // 32, 4, 8).permute(2, 0, 3, 1, 4).contiguous().view(-1)


def from_tiles(t, B, K):
    BT = (B + 15) // 16
    return t.view(K // 32, BT, 4, 16, 8).permute(1, 3, 0, 2, 4).reshape(BT * 16, K)[:B]


@pytest.mark.gpu
@pytest.mark.parametrize("tiled", [0, 1])
@pytest.mark.parametrize("case", [(64, 3840, 1280, 0, 0, 0), (8, 1280, 5120, 0, 1, 0), (33, 5120, 1280, 1, 0, 1),
                                  (128, 8194, 1280, 0, 0, 0), (5, 66, 128, 0, 0, 0), (17, 40, 96, 0, 1, 0)])
def test_skinny_gemm(lib, case, tiled):
    """decode projections at batch > 4: weights streamed once, batch on MFMA; vs fp64 math on the same bf16 inputs.
    tiled: bf16 X (and bf16 Y) in MFMA-fragment order, as the decode step keeps them."""
    B, N, K, gelu, acc, ybf = case
    if tiled and N % 32 and ybf:
        pytest.skip("tiled output needs N % 32 == 0")
    BT = (B + 15) // 16
    x = (rnd("sk.x", (B, K)) * 1.5).to(torch.bfloat16)
    w = (rnd("sk.w", (N, K)) * 0.05).to(torch.bfloat16)
    bias = rnd("sk.b", (N,))
    y0 = rnd("sk.y", (B, N))
    ref = x.double() @ w.double().T + bias.double()
    if gelu:
        ref = 0.5 * ref * (1 + torch.tanh(0.7978845608028654 * (ref + 0.044715 * ref ** 3)))
    if acc:
        ref = ref + y0.double()
    xd, wd, bd = (to_tiles(x, BT) if tiled else x).to(DEV), w.to(DEV), bias.to(DEV)
    layout = 3 if tiled else 0
    y = y0.to(DEV).clone() if not ybf else torch.empty(BT * 16 * N, dtype=torch.bfloat16, device=DEV)
    act = L.ACT_GELU_NEW if gelu else L.ACT_NONE
    L.check(lib.itts_skinny_gemm(y.data_ptr(), ybf, xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), B, N, K, act, acc, 1, None,
                                 layout, stream()))
    torch.cuda.synchronize()
    if ybf:
        got = from_tiles(y.cpu(), B, N) if tiled else y.cpu()[:B * N].view(B, N)
    else:
        got = y.cpu()
    assert relerr(got.float(), ref.float()) < (1e-2 if ybf else 2e-5)
    # rows do not depend on the rest of the batch, nor on the layout (same row alone, row-major, gives the same bits)
    if not acc:
        y1 = torch.empty(16 * N if ybf else N, dtype=y.dtype, device=DEV)
        r = B // 2
        x1 = x[r:r + 1].contiguous().to(DEV)
        L.check(lib.itts_skinny_gemm(y1.data_ptr(), ybf, x1.data_ptr(), wd.data_ptr(), bd.data_ptr(), 1, N, K, act, 0, 1, None, 0,
                                     stream()))
        torch.cuda.synchronize()
        assert torch.equal(y1.cpu()[:N], got[r])
    # the fragment-tiled weight copy (what the engine streams at batch > 4) gives the same bits; its layout is wtile_off
    wt = torch.empty((N + 15) // 16 * 16 * K, dtype=torch.bfloat16, device=DEV)
    L.check(lib.itts_retile_weights(wt.data_ptr(), wd.data_ptr(), N, K, stream()))
    wp = torch.zeros((N + 15) // 16 * 16, K, dtype=torch.bfloat16)
    wp[:N] = w
    assert torch.equal(wt.cpu(), wp.view(-1, 16, K // 32, 4, 8).permute(0, 2, 3, 1, 4).reshape(-1))
    y2 = y0.to(DEV).clone() if not ybf else torch.empty(BT * 16 * N, dtype=torch.bfloat16, device=DEV)
    L.check(lib.itts_skinny_gemm(y2.data_ptr(), ybf, xd.data_ptr(), wt.data_ptr(), bd.data_ptr(), B, N, K, act, acc, 1, None,
                                 layout | 4, stream()))
    torch.cuda.synchronize()
    if ybf and not tiled:
        assert torch.equal(y2.cpu()[:B * N], y.cpu()[:B * N])
    elif ybf:
        assert torch.equal(from_tiles(y2.cpu(), B, N), got)
    else:
        assert torch.equal(y2.cpu(), y.cpu())


@pytest.mark.gpu
@pytest.mark.parametrize("tiledw", [0, 1])
@pytest.mark.parametrize("case", [(6, 3840, 1280, 0, 0), (16, 5120, 1280, 1, 1), (9, 66, 128, 0, 0), (5, 200, 512, 1, 1)])
def test_skinny_gemm_layernorm_prologue(lib, case, tiledw):
    """5-16 decode rows: LayerNorm (affine folded into W) inside the projection that consumes it; vs fp64 LayerNorm of the
    fp32 rows, rounded to bf16 as the row kernel does, times the same bf16 weights."""
    B, N, K, gelu, ybf = case
    x = rnd("skl.x", (B, K)) * 2.5 + 0.3
    w = (rnd("skl.w", (N, K)) * 0.05).to(torch.bfloat16)
    bias = rnd("skl.b", (N,))
    xn = torch.nn.functional.layer_norm(x.double(), (K,), None, None, 1e-5).to(torch.bfloat16)
    ref = xn.double() @ w.double().T + bias.double()
    if gelu:
        ref = 0.5 * ref * (1 + torch.tanh(0.7978845608028654 * (ref + 0.044715 * ref ** 3)))
    xd, wd, bd = x.to(DEV), w.to(DEV), bias.to(DEV)
    if tiledw:
        wt = torch.empty((N + 15) // 16 * 16 * K, dtype=torch.bfloat16, device=DEV)
        L.check(lib.itts_retile_weights(wt.data_ptr(), wd.data_ptr(), N, K, stream()))
        wd = wt
    y = torch.empty(16 * N, dtype=torch.bfloat16, device=DEV) if ybf else torch.empty(B, N, device=DEV)
    act = L.ACT_GELU_NEW if gelu else L.ACT_NONE
    L.check(lib.itts_skinny_gemm(y.data_ptr(), ybf, xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), B, N, K, act, 0, 1, None,
                                 8 | (4 if tiledw else 0), stream()))
    torch.cuda.synchronize()
    got = y.cpu()[:B * N].view(B, N).float() if ybf else y.cpu()
    # a bf16 rounding of LN(x) may land on the neighbouring value where the moments differ in the last ulp
    assert relerr(got, ref.float()) < (1e-2 if ybf else 2e-3)
    # a row does not depend on the rest of the batch
    y1 = torch.empty_like(y)
    r = B // 2
    L.check(lib.itts_skinny_gemm(y1.data_ptr(), ybf, xd[r:r + 1].contiguous().data_ptr(), wd.data_ptr(), bd.data_ptr(), 1, N, K, act, 0,
                                 1, None, 8 | (4 if tiledw else 0), stream()))
    torch.cuda.synchronize()
    assert torch.equal(y1.cpu().view(-1)[:N], y.cpu().view(-1)[r * N:(r + 1) * N])


@pytest.mark.gpu
@pytest.mark.parametrize("tiledw", [0, 1])
@pytest.mark.parametrize("case", [(6, 1280, 5120), (16, 1280, 1280), (11, 72, 96)])
def test_skinny_gemm_half_tiles_accumulate(lib, case, tiledw):
    """5-16 decode rows: the residual projections as 8-feature workgroups, no K split, accumulating into the fp32 stream:
    the same bits as the whole-tile form."""
    B, N, K = case
    x = (rnd("skh.x", (B, K)) * 1.5).to(torch.bfloat16)
    w = (rnd("skh.w", (N, K)) * 0.05).to(torch.bfloat16)
    bias = rnd("skh.b", (N,))
    y0 = rnd("skh.y", (B, N))
    ref = x.double() @ w.double().T + bias.double() + y0.double()
    xd, wd, bd = to_tiles(x, 1).to(DEV), w.to(DEV), bias.to(DEV)
    if tiledw:
        wt = torch.empty((N + 15) // 16 * 16 * K, dtype=torch.bfloat16, device=DEV)
        L.check(lib.itts_retile_weights(wt.data_ptr(), wd.data_ptr(), N, K, stream()))
        wd = wt
    lay = 1 | (4 if tiledw else 0)
    ya, yb = y0.to(DEV).clone(), y0.to(DEV).clone()
    L.check(lib.itts_skinny_gemm(ya.data_ptr(), 0, xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), B, N, K, L.ACT_NONE, 1, 1, None,
                                 lay | 16, stream()))
    L.check(lib.itts_skinny_gemm(yb.data_ptr(), 0, xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), B, N, K, L.ACT_NONE, 1, 1, None,
                                 lay, stream()))
    torch.cuda.synchronize()
    assert relerr(ya.cpu(), ref.float()) < 2e-5
    assert torch.equal(ya.cpu(), yb.cpu())


@pytest.mark.gpu
@pytest.mark.parametrize("passes", [1, 2])
def test_ln_rows_bf16(lib, passes):
    rows, D = 37, 1280
    x = rnd("lnr.x", (rows, D)) * 3 + 0.7
    g, b = rnd("lnr.g", (D,)) * 0.2 + 1, rnd("lnr.b", (D,)) * 0.1
    ref = torch.nn.functional.layer_norm(x.double(), (D,), g.double(), b.double(), 1e-5)
    if passes == 2:
        ref = torch.nn.functional.layer_norm(ref, (D,), None, None, 1e-5)
    xd, gd, bd = x.to(DEV), g.to(DEV), b.to(DEV)
    y = torch.empty(rows, D, dtype=torch.bfloat16, device=DEV)
    L.check(lib.itts_ln_rows_bf16(y.data_ptr(), xd.data_ptr(), gd.data_ptr(), bd.data_ptr(), rows, D, 1e-5, passes, None, 0,
                                  None, 0, stream()))
    yt = torch.empty(((rows + 15) // 16) * 16 * D, dtype=torch.bfloat16, device=DEV)
    L.check(lib.itts_ln_rows_bf16(yt.data_ptr(), xd.data_ptr(), gd.data_ptr(), bd.data_ptr(), rows, D, 1e-5, passes, None, 0,
                                  None, 1, stream()))
    torch.cuda.synchronize()
    assert relerr(y.float(), ref.float()) < 6e-3
    assert torch.equal(from_tiles(yt.cpu(), rows, D), y.cpu())


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(64, 1280, 5120, 4), (64, 1280, 1280, 2), (9, 128, 512, 4), (20, 1280, 1280, 3)])
def test_skinny_splitk_absorbed_by_layernorm(lib, case):
    """residual projection split over K: partial sums + bias are absorbed by the following LayerNorm kernel, which also
    writes the updated fp32 residual stream (two-stage deterministic reduction)."""
    B, N, K, S = case
    x = (rnd("ss.x", (B, K)) * 1.5).to(torch.bfloat16)
    w = (rnd("ss.w", (N, K)) * 0.05).to(torch.bfloat16)
    bias = rnd("ss.b", (N,))
    h0 = rnd("ss.h", (B, N)) * 2
    h_ref = h0.double() + x.double() @ w.double().T + bias.double()
    y_ref = torch.nn.functional.layer_norm(h_ref, (N,), None, None, 1e-5)
    xd, wd, bd, hd = x.to(DEV), w.to(DEV), bias.to(DEV), h0.to(DEV).clone()
    part = torch.full((S, B, N), float("nan"), device=DEV)
    L.check(lib.itts_skinny_gemm(None, 0, xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), B, N, K, L.ACT_NONE, 1, S,
                                 part.data_ptr(), 0, stream()))
    y = torch.empty(B, N, dtype=torch.bfloat16, device=DEV)
    L.check(lib.itts_ln_rows_bf16(y.data_ptr(), hd.data_ptr(), None, None, B, N, 1e-5, 1, part.data_ptr(), S, bd.data_ptr(),
                                  0, stream()))
    torch.cuda.synchronize()
    assert relerr(hd, h_ref.float()) < 2e-5
    assert relerr(y.float(), y_ref.float()) < 6e-3
    # run-to-run determinism
    hd2 = h0.to(DEV).clone()
    L.check(lib.itts_skinny_gemm(None, 0, xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), B, N, K, L.ACT_NONE, 1, S,
                                 part.data_ptr(), 0, stream()))
    L.check(lib.itts_ln_rows_bf16(y.data_ptr(), hd2.data_ptr(), None, None, B, N, 1e-5, 1, part.data_ptr(), S, bd.data_ptr(),
                                  0, stream()))
    torch.cuda.synchronize()
    assert torch.equal(hd, hd2)


AMP_CASES = [
    # B, T, C, k, dil   (AMPBlock1 convs of the narrow BigVGAN stages; T not a multiple of the time tile)
    (2, 700, 24, 3, 1), (1, 1000, 24, 11, 5), (2, 515, 48, 7, 3), (1, 769, 48, 11, 1), (2, 300, 96, 3, 5), (1, 641, 96, 11, 5),
    (1, 256, 96, 7, 1), (3, 257, 24, 7, 5),
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", AMP_CASES)
def test_conv_lds_amp_shapes(lib, case):
    """LDS-resident conv kernel (conv_lds.hip) vs fp64 torch conv1d on the same bf16 inputs, with the full AMP epilogue:
    per-batch bias, residual, alpha scaling and beta * running-sum accumulate; also bit-compared per row against the
    generic shift-GEMM path within bf16 rounding."""
    B, T, Cc, k, dil = case
    x = rnd(f"al.x{case}", (B, Cc, T)).to(torch.bfloat16)
    w = rnd(f"al.w{case}", (Cc, Cc, k), 1.0 / np.sqrt(Cc * k)).to(torch.bfloat16)
    bias = rnd(f"al.b{case}", (B, Cc), 0.1)
    res = rnd(f"al.r{case}", (B, Cc, T)).to(torch.bfloat16)
    run = rnd(f"al.s{case}", (B, Cc, T)).to(torch.bfloat16)
    pad = dil * (k - 1) // 2
    ref = F.conv1d(F.pad(x.double(), (pad, pad)), w.double(), None, dilation=dil) + bias.double()[:, :, None]
    ref = (ref + res.double()) * (1.0 / 3.0) + 1.0 * run.double()
    A = x.transpose(1, 2).contiguous().to(DEV)
    W = torch.from_numpy(pack.conv_w(w.float().numpy())).to(torch.bfloat16).to(DEV)
    R = res.transpose(1, 2).contiguous().to(DEV)
    S = run.transpose(1, 2).contiguous().to(DEV)
    kw = dict(M=B * T, N=Cc, Cin=Cc, taps=k, lda=Cc, ldc=Cc, T=T, dil=dil, pad_left=pad, pad_mode=0, bias=bias.to(DEV),
              bias_bstride=Cc, R=R, ldr=Cc, alpha=1.0 / 3.0, ADD=S, ldadd=Cc, beta=1.0)
    out = run_gemm(lib, A, W, (B, T, Cc), L.BF16, L.BF16, L.BF16, 0, **kw)
    assert relerr(out.float().transpose(1, 2), ref.float()) < 1.5e-2
    gen = run_gemm(lib, A, W, (B, T, Cc), L.BF16, L.BF16, L.BF16, 1, **kw)  # vector kernel, fp32 accumulate
    assert relerr(out.float(), gen.float()) < 1.5e-2


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(1, 8, 255, 255), (2, 3, 70, 133)])
def test_attention_mfma_wide_keys(lib, case):
    """conformer rel-pos attention shape: 128-wide queries / keys ([q+u | q+v] . [k | p]), 64-wide values, non-causal."""
    B, H, Sq, Sk = case
    q = rnd("aw.q", (B, Sq, H * 128)).to(torch.bfloat16)
    k = rnd("aw.k", (B, Sk, H * 128)).to(torch.bfloat16)
    v = rnd("aw.v", (B, Sk, H * 64)).to(torch.bfloat16)
    qf = q.double().view(B, Sq, H, 128).transpose(1, 2)
    kf = k.double().view(B, Sk, H, 128).transpose(1, 2)
    vf = v.double().view(B, Sk, H, 64).transpose(1, 2)
    ref = (torch.softmax(qf @ kf.transpose(-1, -2) * 0.125, -1) @ vf).transpose(1, 2).reshape(B, Sq, H * 64)
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    o = torch.full((B, Sq, H * 64), float("nan"), dtype=torch.bfloat16, device=DEV)
    L.check(lib.itts_attention(o.data_ptr(), qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), B, H, Sq, Sk, 128, 64, H * 128,
                               H * 128, H * 64, H * 64, 0.125, 0, None, L.BF16, stream()))
    torch.cuda.synchronize()
    assert relerr(o.float(), ref.float()) < 1.5e-2


# ---- LDS-DMA staged GEMM (gemm_glds.hip): the large regular conv shapes of BigVGAN stages 1-3 ---------------------
GLDS_CASES = [
    # B, T, Cin, Cout, k, dil, mode        (tiles >= 384 and K >= 4096 or N = 192, so the dispatcher takes the glds kernel)
    (2, 8192, 192, 192, 7, 3, "zeros"),    # BN = 64 tiles (N = 192), dilated taps, zero padding at both sequence ends
    (3, 5461, 384, 384, 11, 1, "zeros"),   # BN = 128 (K = 4224), M not a multiple of 128, batch boundaries inside tiles
    (2, 8200, 384, 256, 11, 5, "reflect"),  # reflect padding resolved in the DMA source address
    (1, 16384, 4096, 1280, 1, 1, "zeros"),  # plain linear, K = 4096
]


@pytest.mark.parametrize("case", GLDS_CASES)
@pytest.mark.parametrize("out_f32", [False, True])
def test_gemm_glds_conv(lib, case, out_f32):
    B, T, Cin, Cout, k, dil, mode = case
    x = rnd(f"gl.x{case}", (B, Cin, T)).to(torch.bfloat16)
    w = rnd(f"gl.w{case}", (Cout, Cin, k), 1.0 / np.sqrt(Cin * k)).to(torch.bfloat16)
    bias = rnd(f"gl.b{case}", (B, Cout), 0.1)
    pad = dil * (k - 1) // 2
    xp = F.pad(x.float(), (pad, pad), mode="reflect") if mode == "reflect" else F.pad(x.float(), (pad, pad))
    ref = F.conv1d(xp, w.float(), None, dilation=dil) + bias[:, :, None]
    res = rnd(f"gl.r{case}", (B, Cout, T)).to(torch.float32 if out_f32 else torch.bfloat16)
    add = rnd(f"gl.a{case}", (B, Cout, T)).to(torch.float32 if out_f32 else torch.bfloat16)
    ref = (ref + res.float()) * (1.0 / 3.0) + 0.5 * add.float()
    A = x.transpose(1, 2).contiguous().to(DEV)
    W = torch.from_numpy(pack.conv_w(w.float().numpy())).to(torch.bfloat16).to(DEV)
    R, ADD = res.transpose(1, 2).contiguous().to(DEV), add.transpose(1, 2).contiguous().to(DEV)
    outs = []
    for env in (None, "1"):  # the glds kernel, then the register-staged one on the same arguments
        import os

        if env:
            os.environ["ITTS_GEMM_FORCE_OLD"] = env
        try:
            outs.append(run_gemm(lib, A, W, (B, T, Cout), L.BF16, L.BF16, L.F32 if out_f32 else L.BF16, 0, M=B * T, N=Cout, Cin=Cin,
                                 taps=k, lda=Cin, ldc=Cout, T=T, dil=dil, pad_left=pad, pad_mode=1 if mode == "reflect" else 0,
                                 bias=bias.to(DEV), bias_bstride=Cout, R=R, ldr=Cout, alpha=1.0 / 3.0, ADD=ADD, ldadd=Cout, beta=0.5))
        finally:
            os.environ.pop("ITTS_GEMM_FORCE_OLD", None)
    assert relerr(outs[0].float().transpose(1, 2), ref) < (3e-3 if out_f32 else 2e-2)
    # both kernels accumulate the same bf16 products in fp32: they agree far inside the bf16 output rounding
    assert relerr(outs[0].float(), outs[1].float()) < (1e-4 if out_f32 else 1e-2)


def test_gemm_glds_transposed_conv(lib):
    B, T, Cin, Cout, k, u = 2, 4096, 768, 384, 8, 4
    p = (k - u) // 2
    x = rnd("glt.x", (B, Cin, T)).to(torch.bfloat16)
    w = rnd("glt.w", (Cin, Cout, k), 1.0 / np.sqrt(Cin * k / u)).to(torch.bfloat16)
    bias = rnd("glt.b", (B, Cout), 0.1)
    ref = F.conv_transpose1d(x.float(), w.float(), None, stride=u, padding=p) + bias[:, :, None]
    A = x.transpose(1, 2).contiguous().to(DEV)
    W = torch.from_numpy(pack.convT_w(w.float().numpy(), u, p)).to(torch.bfloat16).to(DEV)
    out = run_gemm(lib, A, W, (B, T * u, Cout), L.BF16, L.BF16, L.BF16, 0, M=B * T, N=Cout, Cin=Cin, taps=k // u, lda=Cin,
                   ldc=u * Cout, T=T, dil=-1, pad_left=0, nphase=u, phase_shift=[(ph + p) // u for ph in range(u)],
                   bias=bias.to(DEV), bias_bstride=Cout)
    assert relerr(out.float().transpose(1, 2), ref) < 2e-2


# ---- 256 x 256 eight-phase GEMM (gemm_p8.hip): the large regular shapes at batch (BigVGAN C >= 384 convs, conv_pre, GPT projections) ---
P8_CASES = [
    # B, T, Cin, Cout, k, dil, mode       (>= 384 tiles of 256 x 256, so the dispatcher takes the eight-phase kernel)
    (2, 16384, 384, 768, 7, 3, "zeros"),    # 42 K-tiles, dilated taps, zero padding at both ends of both batch items
    (2, 24576, 192, 512, 3, 5, "reflect"),  # 9 K-tiles (ODD: the pipeline's overshoot stages), reflect padding in the DMA source
    (3, 22000, 384, 384, 3, 1, "zeros"),    # N = 384: the 512 x 128 geometry; M not a multiple of 512, item boundaries inside tiles
    (2, 24576, 256, 640, 11, 5, "reflect"),  # N = 640 = 5 x 128 (512 x 128 tiles), 44 K-tiles, reflect
    (1, 32768, 1280, 3840, 1, 1, "zeros"),  # plain linear (GPT c_attn at batch): 20 K-tiles, 15 column tiles
]


@pytest.mark.parametrize("case", P8_CASES)
@pytest.mark.parametrize("out_f32", [False, True])
def test_gemm_p8_conv(lib, case, out_f32):
    import os

    B, T, Cin, Cout, k, dil, mode = case
    x = rnd(f"p8.x{case}", (B, Cin, T)).to(torch.bfloat16)
    w = rnd(f"p8.w{case}", (Cout, Cin, k), 1.0 / np.sqrt(Cin * k)).to(torch.bfloat16)
    bias = rnd(f"p8.b{case}", (B, Cout), 0.1)
    pad = dil * (k - 1) // 2
    xp = F.pad(x.float(), (pad, pad), mode="reflect") if mode == "reflect" else F.pad(x.float(), (pad, pad))
    ref = F.conv1d(xp, w.float(), None, dilation=dil) + bias[:, :, None]
    res = rnd(f"p8.r{case}", (B, Cout, T)).to(torch.float32 if out_f32 else torch.bfloat16)
    add = rnd(f"p8.a{case}", (B, Cout, T)).to(torch.float32 if out_f32 else torch.bfloat16)
    ref = (ref + res.float()) * (1.0 / 3.0) + 0.5 * add.float()
    A = x.transpose(1, 2).contiguous().to(DEV)
    W = torch.from_numpy(pack.conv_w(w.float().numpy())).to(torch.bfloat16).to(DEV)
    R, ADD = res.transpose(1, 2).contiguous().to(DEV), add.transpose(1, 2).contiguous().to(DEV)
    outs, which = [], []
    for env in (None, "1"):  # the eight-phase kernel, then the register-staged one on the same arguments
        if env:
            os.environ["ITTS_GEMM_FORCE_OLD"] = env
        try:
            outs.append(run_gemm(lib, A, W, (B, T, Cout), L.BF16, L.BF16, L.F32 if out_f32 else L.BF16, 0, M=B * T, N=Cout, Cin=Cin,
                                 taps=k, lda=Cin, ldc=Cout, T=T, dil=dil, pad_left=pad, pad_mode=1 if mode == "reflect" else 0,
                                 bias=bias.to(DEV), bias_bstride=Cout, R=R, ldr=Cout, alpha=1.0 / 3.0, ADD=ADD, ldadd=Cout, beta=0.5))
            which.append(run_gemm.which)
        finally:
            os.environ.pop("ITTS_GEMM_FORCE_OLD", None)
    assert which == [3, 1], which
    assert relerr(outs[0].float().transpose(1, 2), ref) < (3e-3 if out_f32 else 2e-2)
    # both kernels accumulate the same bf16 products in fp32: they agree far inside the bf16 output rounding
    assert relerr(outs[0].float(), outs[1].float()) < (1e-4 if out_f32 else 1e-2)
    # a staging race would show as rare wrong tiles: the WORST element, not only the norm, against the other kernel
    d = (outs[0].float() - outs[1].float()).abs().max().item()
    assert d < (2e-3 if out_f32 else 6e-2), d


def test_gemm_p8_repeats_are_identical(lib):
    """The pipeline keeps three chunks in flight across every barrier: a missing wait shows as run-to-run differences."""
    B, T, Cin, Cout, k = 2, 16384, 768, 768, 3
    x = rnd("p8r.x", (B, T, Cin)).to(torch.bfloat16).to(DEV)
    W = torch.from_numpy(pack.conv_w(rnd("p8r.w", (Cout, Cin, k), 1.0 / np.sqrt(Cin * k)).float().numpy())).to(torch.bfloat16).to(DEV)
    outs = [run_gemm(lib, x, W, (B, T, Cout), L.BF16, L.BF16, L.BF16, 0, M=B * T, N=Cout, Cin=Cin, taps=k, lda=Cin, ldc=Cout, T=T,
                     dil=1, pad_left=1) for _ in range(6)]
    assert run_gemm.which == 3
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


KSPLIT_CASES = [
    # B, T, Cin, Cout, k, dil, mode, forced split (None = the planner's choice)
    (1, 1242, 5120, 1280, 1, 1, "zeros", None),   # batch-1 latent mlp.c_proj: 25 tiles, 80 K-tiles -> 8 splits of 10
    (2, 480, 1280, 1536, 7, 1, "zeros", None),    # BigVGAN conv_pre at batch 1: 24 tiles, 140 K-tiles (8 splits of 18, the last of 14)
    (2, 1920, 768, 768, 11, 5, "reflect", None),  # stage-0 AMP conv: 45 tiles, 132 K-tiles, splits start inside a tap (12 chunks per tap)
    (2, 1920, 768, 768, 7, 3, "zeros", 5),        # 84 K-tiles in 5 splits of 17: the last split has 16, odd counts (overshoot stages)
    (1, 1000, 384, 384, 3, 1, "zeros", 3),        # 512 x 128 geometry: 2 x 3 tiles, 18 K-tiles in 3 splits of 6; M not a multiple of 512
    (1, 142, 5120, 1280, 1, 1, "zeros", None),    # one-sentence prefill mlp.c_proj: ONE row tile, 142 of its 256 rows exist; 8 splits
    (20, 142, 5120, 1280, 1, 1, "zeros", None),   # 20-sentence prefill: 60 tiles -> 3 splits of 27 / 27 / 26
]


@pytest.mark.parametrize("case", KSPLIT_CASES)
@pytest.mark.parametrize("out_f32", [False, True])
def test_gemm_p8_ksplit(lib, case, out_f32):
    """K split over workgroups (itts_gemm_ws / the engine's few-tile GEMMs): raw sums per split, second launch adds them in split order
    and runs the epilogue - against torch, against the register-staged kernel on the same arguments, and bit-identical on repeats."""
    import os

    B, T, Cin, Cout, k, dil, mode, force = case
    x = rnd(f"ks.x{case}", (B, Cin, T)).to(torch.bfloat16)
    w = rnd(f"ks.w{case}", (Cout, Cin, k), 1.0 / np.sqrt(Cin * k)).to(torch.bfloat16)
    bias = rnd(f"ks.b{case}", (B, Cout), 0.1)
    pad = dil * (k - 1) // 2
    xp = F.pad(x.float(), (pad, pad), mode="reflect") if mode == "reflect" else F.pad(x.float(), (pad, pad))
    ref = F.gelu(F.conv1d(xp, w.float(), None, dilation=dil) + bias[:, :, None], approximate="tanh")
    res = rnd(f"ks.r{case}", (B, Cout, T)).to(torch.float32 if out_f32 else torch.bfloat16)
    ref = (ref + res.float()) * 0.5
    A = x.transpose(1, 2).contiguous().to(DEV)
    W = torch.from_numpy(pack.conv_w(w.float().numpy())).to(torch.bfloat16).to(DEV)
    R = res.transpose(1, 2).contiguous().to(DEV)
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
    kw = dict(M=B * T, N=Cout, Cin=Cin, taps=k, lda=Cin, ldc=Cout, T=T, dil=dil, pad_left=pad, pad_mode=1 if mode == "reflect" else 0,
              bias=bias.to(DEV), bias_bstride=Cout, act=3, R=R, ldr=Cout, alpha=0.5)
    dt_c = L.F32 if out_f32 else L.BF16
    if force:
        os.environ["ITTS_GEMM_KSPLIT"] = str(force)
    try:
        outs = [run_gemm(lib, A, W, (B, T, Cout), L.BF16, L.BF16, dt_c, 0, ws=ws, **kw) for _ in range(3)]
        nsplit = run_gemm.ksplit
    finally:
        os.environ.pop("ITTS_GEMM_KSPLIT", None)
    assert nsplit == (force or nsplit) and nsplit >= 2, nsplit
    plain = run_gemm(lib, A, W, (B, T, Cout), L.BF16, L.BF16, dt_c, 0, **kw)  # no workspace: the unsplit dispatch
    assert run_gemm.which in (1, 2, 3)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert relerr(outs[0].float().transpose(1, 2), ref) < (3e-3 if out_f32 else 2e-2)
    assert relerr(outs[0].float(), plain.float()) < (1e-4 if out_f32 else 1e-2)
    d = (outs[0].float() - plain.float()).abs().max().item()
    assert d < (2e-3 if out_f32 else 6e-2), d


def test_gemm_ws_leaves_other_shapes_alone(lib):
    """itts_gemm_ws on a shape with plenty of tiles (or too little K) is itts_gemm: same kernel, same bits."""
    B, T, Cin, Cout, k = 2, 16384, 768, 768, 3
    x = rnd("p8r.x", (B, T, Cin)).to(torch.bfloat16).to(DEV)
    W = torch.from_numpy(pack.conv_w(rnd("p8r.w", (Cout, Cin, k), 1.0 / np.sqrt(Cin * k)).float().numpy())).to(torch.bfloat16).to(DEV)
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
    kw = dict(M=B * T, N=Cout, Cin=Cin, taps=k, lda=Cin, ldc=Cout, T=T, dil=1, pad_left=1)
    a = run_gemm(lib, x, W, (B, T, Cout), L.BF16, L.BF16, L.BF16, 0, ws=ws, **kw)
    assert run_gemm.ksplit == 1
    b = run_gemm(lib, x, W, (B, T, Cout), L.BF16, L.BF16, L.BF16, 0, **kw)
    assert torch.equal(a, b)


@pytest.mark.parametrize("dims", [(2, 6144, 1536, 768), (2, 12288, 768, 384)])
def test_gemm_p8_transposed_conv(lib, dims):
    """ConvTranspose1d(k = 8, stride 4) as four polyphase column-tile groups of the eight-phase kernel (BigVGAN ups 0 / 1 at batch:
    N = 768 on 256 x 256 tiles, N = 384 on 512 x 128 tiles), against torch and against the register-staged kernel."""
    import os

    B, T, Cin, Cout = dims
    k, u = 8, 4
    p = (k - u) // 2
    x = rnd(f"p8t.x{dims}", (B, Cin, T)).to(torch.bfloat16)
    w = rnd(f"p8t.w{dims}", (Cin, Cout, k), 1.0 / np.sqrt(Cin * k / u)).to(torch.bfloat16)
    bias = rnd(f"p8t.b{dims}", (B, Cout), 0.1)
    ref = F.conv_transpose1d(x.float(), w.float(), None, stride=u, padding=p) + bias[:, :, None]
    A = x.transpose(1, 2).contiguous().to(DEV)
    W = torch.from_numpy(pack.convT_w(w.float().numpy(), u, p)).to(torch.bfloat16).to(DEV)
    outs, which = [], []
    for env in (None, "1"):
        if env:
            os.environ["ITTS_GEMM_FORCE_OLD"] = env
        try:
            outs.append(run_gemm(lib, A, W, (B, T * u, Cout), L.BF16, L.BF16, L.BF16, 0, M=B * T, N=Cout, Cin=Cin, taps=k // u, lda=Cin,
                                 ldc=u * Cout, T=T, dil=-1, pad_left=0, nphase=u, phase_shift=[(ph + p) // u for ph in range(u)],
                                 bias=bias.to(DEV), bias_bstride=Cout))
            which.append(run_gemm.which)
        finally:
            os.environ.pop("ITTS_GEMM_FORCE_OLD", None)
    assert which == [3, 1], which
    assert relerr(outs[0].float().transpose(1, 2), ref) < 2e-2
    assert relerr(outs[0].float(), outs[1].float()) < 1e-2
