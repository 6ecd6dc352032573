"""Thin torch-tensor wrappers over the kernel-level C entry points (used by tests, the eval harness and bench)."""
import ctypes as C

import torch

from . import lib as L


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


PREC = {None: 0, "f32": 1, "f16x3": 2, "bf16": 3}


def split16(w, scale=1.0):
    """fp16-split image of (an f32 cuda tensor * scale), numel % 4 == 0: what `gemm(..., prec="f16x3", w_split=scale)`
    consumes; scale is a power of two."""
    lib = L.load()
    w = w.contiguous()
    out = torch.empty_like(w)
    L.check(lib.actmi_op_split16(_p(w), _p(out), w.numel(), float(scale), L.current_stream_ptr()), None, "op_split16")
    return out


def permute_conv_k(w_rows, taps, cin):
    """[rows, ld] convolution weight rows, the first taps*cin columns re-ordered (tap, c) -> (c / 32, tap, c % 32): the K order
    the engine's split images use (actmi_gemm_desc.k_tap_inner)."""
    lib = L.load()
    w_rows = w_rows.contiguous()
    out = torch.empty_like(w_rows)
    rows, ld = w_rows.numel() // w_rows.shape[-1], w_rows.shape[-1]
    L.check(lib.actmi_op_permute_conv_k(_p(w_rows), _p(out), rows, int(taps), int(cin), int(ld), L.current_stream_ptr()), None,
            "op_permute_conv_k")
    return out


def pow2_scale(x):
    """[scale, scratch]: scale = the power of two that brings max|x| into [2^13, 2^14) (device-side, no host sync);
    pass the tensor as gemm(..., a_scale_dev=) / b_scale_dev= for operands far from the fp16 range."""
    lib = L.load()
    assert x.dim() == 2 and x.stride(1) == 1
    out = torch.zeros(2, dtype=torch.float32, device=x.device)
    L.check(lib.actmi_op_pow2_scale(_p(x), x.stride(0), x.shape[0], x.shape[1], _p(out), L.current_stream_ptr()), None,
            "op_pow2_scale")
    return out


def auto_splitk(M, N, K):
    """Split factor for a product whose grid is too small to fill 256 CUs while every workgroup walks a long contraction (the
    engine's rule for its own small-batch launches, engine.hip:ctx_gemm): aim at 1536 64x64-tile equivalents, keep >= 12 K
    tiles of 32 per split, at most 8 splits; 0 = do not split."""
    tiles = ((M + 63) // 64) * ((N + 63) // 64)
    nk = (K + 31) // 32
    if tiles >= 768 or nk < 24:
        return 0
    s = min(8, (1536 + tiles - 1) // tiles, nk // 12)
    while s >= 2 and (s - 1) * ((nk + s - 1) // s) >= nk:
        s -= 1
    return s if s >= 2 else 0


def gemm(A, W, bias=None, scale=None, res=None, res_mod=0, relu=False, a_add=None, add_mod=0, add_ncols=0, rowmap=None,
         out=None, out_rows=None, drop_p=0.0, drop_seed=0, prec=None, w_split=False, a_scale=0.0, b_scale=0.0,
         a_scale_dev=None, b_scale_dev=None, splitk=0):
    """out[rowmap(m)] = act((A' @ W.T) * scale + bias + res[m % res_mod]); A [M,K], W [N,K] row-major f32 cuda.
    splitk > 1: the contraction is split into that many plain slices which a combine pass sums in a fixed order before
    the epilogue (what the engine does for small grids); no rowmap / res_mod / dropout in that form."""
    lib = L.load()
    M, K = A.shape
    N = W.shape[0]
    if splitk == "auto":
        splitk = auto_splitk(M, N, K) if (rowmap is None and not res_mod and not drop_p and N % 4 == 0 and a_scale_dev is None
                                          and b_scale_dev is None) else 0
    if splitk and splitk > 1:
        assert rowmap is None and not res_mod and not drop_p
        part = torch.empty((splitk, M, N), dtype=torch.float32, device=A.device)
        d = L.GemmDesc()
        d.A, d.lda, d.mode = A.data_ptr(), A.stride(0), 0
        if a_add is not None:
            d.A_add, d.ld_add, d.add_mod, d.add_ncols = a_add.data_ptr(), a_add.stride(0), add_mod, add_ncols
        d.Bw, d.ldb = W.data_ptr(), W.stride(0)
        d.C, d.ldc = part.data_ptr(), N
        d.M, d.N, d.K, d.groups = M, N, K, 1
        d.splitk, d.split_stride = int(splitk), M * N
        d.prec, d.b_split, d.b_scale = PREC[prec], 1 if w_split else 0, float(w_split) if w_split else float(b_scale)
        d.a_scale = float(a_scale)
        L.check(lib.actmi_op_gemm(C.byref(d), L.current_stream_ptr()), None, "op_gemm")
        if out is None:
            out = torch.empty((M, N), dtype=torch.float32, device=A.device)
        L.check(lib.actmi_op_splitk_combine(part.data_ptr(), int(splitk), M * N, N, M, N, _p(scale), _p(bias), _p(res),
                                            res.stride(0) if res is not None else 0,
                                            2 if relu == "gelu" else (1 if relu else 0), out.data_ptr(), out.stride(0),
                                            L.current_stream_ptr()), None, "op_splitk_combine")
        return out
    if out is None:
        out = torch.zeros((out_rows or M, N), dtype=torch.float32, device=A.device)
    d = L.GemmDesc()
    d.A, d.lda, d.mode = A.data_ptr(), A.stride(0), 0
    if a_add is not None:
        d.A_add, d.ld_add, d.add_mod, d.add_ncols = a_add.data_ptr(), a_add.stride(0), add_mod, add_ncols
    d.Bw, d.ldb = W.data_ptr(), W.stride(0)
    d.scale = scale.data_ptr() if scale is not None else None
    d.bias = bias.data_ptr() if bias is not None else None
    if res is not None:
        d.res, d.ldres, d.res_mod = res.data_ptr(), res.stride(0), res_mod
    d.relu = 2 if relu == "gelu" else (1 if relu else 0)
    d.C, d.ldc = out.data_ptr(), out.stride(0)
    d.rowmap = rowmap.data_ptr() if rowmap is not None else None
    d.M, d.N, d.K, d.groups = M, N, K, 1
    d.drop_p, d.drop_seed = float(drop_p), int(drop_seed)
    d.prec, d.b_split, d.b_scale = PREC[prec], 1 if w_split else 0, float(w_split) if w_split else float(b_scale)
    d.a_scale = float(a_scale)
    d.a_scale_dev = a_scale_dev.data_ptr() if a_scale_dev is not None else None
    d.b_scale_dev = b_scale_dev.data_ptr() if b_scale_dev is not None else None
    L.check(lib.actmi_op_gemm(C.byref(d), L.current_stream_ptr()), None, "op_gemm")
    return out


def conv2d_with_second_source(y1, x, wf_split, w_scale, bias, stride_x=2, relu=True, splitk=0, k_tap_inner=False):
    """A ResNet block's conv2 with the block's 1x1 / stride-2 downsample branch in the same contraction (gemm.hip second
    source): y1 [G,B,H,W,C] is convolved 3x3 / s1 / p1, x [G,B,Hx,Wx,Cx] joins at stride_x as extra columns of the contraction.
    wf_split: split16 image (built with w_scale) of [G][Cout][9*C + Cx]; bias [G,Cout].  f16x3 only.  Returns [G,B,H,W,Cout]."""
    lib = L.load()
    G, B, H, W, Cc = y1.shape
    _, _, Hx, Wx, Cx = x.shape
    Cout = wf_split.shape[1]
    Kf = 9 * Cc + Cx
    assert wf_split.numel() == G * Cout * Kf
    M = B * H * W

    def desc(C_ptr, ldc, gC):
        d = L.GemmDesc()
        d.A, d.mode = y1.data_ptr(), 1
        d.H, d.W, d.Cin, d.KH, d.KW, d.stride, d.pad, d.Ho, d.Wo = H, W, Cc, 3, 3, 1, 1, H, W
        d.img_stride = H * W * Cc
        d.Ax, d.kx_begin, d.Hx, d.Wx, d.Cx, d.stride_x, d.gAx = x.data_ptr(), 9 * Cc, Hx, Wx, Cx, stride_x, B * Hx * Wx * Cx
        d.Bw, d.ldb = wf_split.data_ptr(), Kf
        d.C, d.ldc = C_ptr, ldc
        d.M, d.N, d.K, d.groups = M, Cout, Kf, G
        d.gA, d.gB, d.gSB, d.gC = B * H * W * Cc, Cout * Kf, Cout, gC
        d.prec, d.b_split, d.b_scale = PREC["f16x3"], 1, float(w_scale)
        d.k_tap_inner = 1 if k_tap_inner else 0
        return d
    out = torch.empty((G, B, H, W, Cout), dtype=torch.float32, device=y1.device)
    if splitk and splitk > 1:
        part = torch.empty((G, splitk, M, Cout), dtype=torch.float32, device=y1.device)
        d = desc(part.data_ptr(), Cout, splitk * M * Cout)
        d.splitk, d.split_stride = int(splitk), M * Cout
        L.check(lib.actmi_op_gemm(C.byref(d), L.current_stream_ptr()), None, "op_gemm(conv + second source, split)")
        for g in range(G):
            L.check(lib.actmi_op_splitk_combine(part[g].data_ptr(), int(splitk), M * Cout, Cout, M, Cout, None, _p(bias[g]), None, 0,
                                                1 if relu else 0, out[g].data_ptr(), Cout, L.current_stream_ptr()), None, "combine")
        return out
    d = desc(out.data_ptr(), Cout, M * Cout)
    d.bias = bias.data_ptr()
    d.relu = 1 if relu else 0
    L.check(lib.actmi_op_gemm(C.byref(d), L.current_stream_ptr()), None, "op_gemm(conv + second source)")
    return out


def conv2d_nhwc(x, w_ohwi, scale=None, bias=None, res=None, relu=False, stride=1, pad=1, prec=None, w_split=False, b_scale=0.0,
                k_tap_inner=False):
    """x [G,B,H,W,Cin] camera-major NHWC; w_ohwi [G,Cout,KH,KW,Cin]; scale/bias [G,Cout]; returns [G,B,Ho,Wo,Cout].
    k_tap_inner: the rows of w_ohwi were re-ordered by permute_conv_k (channel blocks outer, taps inner)."""
    lib = L.load()
    G, B, H, W, Cin = x.shape
    _, Cout, KH, KW, _ = w_ohwi.shape
    Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    out = torch.empty((G, B, Ho, Wo, Cout), dtype=torch.float32, device=x.device)     # every element is written (no split-K here)
    d = L.GemmDesc()
    d.A, d.mode = x.data_ptr(), 1
    d.H, d.W, d.Cin, d.KH, d.KW, d.stride, d.pad, d.Ho, d.Wo = H, W, Cin, KH, KW, stride, pad, Ho, Wo
    d.img_stride = H * W * Cin
    d.Bw, d.ldb = w_ohwi.data_ptr(), KH * KW * Cin
    d.scale = scale.data_ptr() if scale is not None else None
    d.bias = bias.data_ptr() if bias is not None else None
    if res is not None:
        d.res, d.ldres = res.data_ptr(), Cout
    d.relu = 1 if relu else 0
    d.C, d.ldc = out.data_ptr(), Cout
    d.M, d.N, d.K, d.groups = B * Ho * Wo, Cout, KH * KW * Cin, G
    d.gA, d.gB, d.gSB = B * H * W * Cin, Cout * KH * KW * Cin, Cout
    d.gC = d.gRes = B * Ho * Wo * Cout
    # f16x3: w_split = scale of a pre-split weight image; b_scale = power-of-two scale applied to plain fp32 weights on the fly
    d.prec, d.b_split, d.b_scale = PREC[prec], 1 if w_split else 0, float(w_split) if w_split else float(b_scale)
    d.k_tap_inner = 1 if k_tap_inner else 0
    L.check(lib.actmi_op_gemm(C.byref(d), L.current_stream_ptr()), None, "op_gemm(conv)")
    return out


def sample_onehot(logits, temperature=1.0, seed=0, want_probs=False):
    """one categorical draw per row of logits [n, V] -> one-hot [n, V] (device-side inverse CDF, counter-based generator)."""
    lib = L.load()
    logits = logits.contiguous()
    n, V = logits.shape
    code = torch.empty_like(logits)
    probs = torch.empty_like(logits) if want_probs else None
    L.check(lib.actmi_op_sample_onehot(_p(logits), n, V, float(temperature), C.c_uint64(int(seed)), _p(probs), _p(code),
                                       L.current_stream_ptr()), None, "op_sample_onehot")
    return (code, probs) if want_probs else code


def conv1_prepare(w_oihw, lut_mode=1):
    """prepared stem weights (actmi_op_conv1_prepare): w_oihw [C, Cout, 3, 7, 7]; lut_mode 0 = the ACT path's ImageNet
    normalisation of the u8 pixels, 1 = v / 255 only.  Returns the workspace tensor to hand to conv1_prepared."""
    lib = L.load()
    w_oihw = w_oihw.contiguous()
    Cc, Cout = w_oihw.shape[0], w_oihw.shape[1]
    ws = torch.zeros(int(lib.actmi_op_conv1_workspace_floats(Cc, Cout)), dtype=torch.float32, device=w_oihw.device)
    L.check(lib.actmi_op_conv1_prepare(_p(w_oihw), _p(ws), Cc, Cout, int(lut_mode), L.current_stream_ptr()), None, "op_conv1_prepare")
    return ws


def conv1_prepared(image_u8, ws, Cout, relu=False, scale=None, bias=None):
    """the 7x7 / s2 stem on prepared weights: image u8 [B, C, H, W, 3] -> [C, B, Ho, Wo, Cout] (f16x3); launch only."""
    lib = L.load()
    image_u8 = image_u8.contiguous()
    B, Cc, H, W, _ = image_u8.shape
    out = torch.empty((Cc, B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, Cout), dtype=torch.float32, device=image_u8.device)
    L.check(lib.actmi_op_conv1_prepared(_p(image_u8), _p(ws), _p(scale), _p(bias), _p(out), B, Cc, H, W, Cout, 1 if relu else 0,
                                        L.current_stream_ptr()), None, "op_conv1_prepared")
    return out


def conv3x3_c64(x, w_ohwi, scale=None, bias=None, res=None, relu=False, w_scale=256.0, w16=None):
    """direct 3x3/s1/p1 conv, 64 -> 64 channels, f16x3: x [G,B,H,W,64]; w_ohwi [G,64,3,3,64]; returns [G,B,H,W,64].
    w16: the split image of w_ohwi built with w_scale (split16) when the caller keeps one; else it is built per call."""
    lib = L.load()
    G, B, H, W, Cin = x.shape
    assert Cin == 64 and tuple(w_ohwi.shape[1:]) == (64, 3, 3, 64)
    out = torch.empty_like(x)
    if w16 is None:
        w16 = split16(w_ohwi, w_scale)
    scale = scale if scale is not None else torch.ones(G, 64, device=x.device)
    bias = bias if bias is not None else torch.zeros(G, 64, device=x.device)
    L.check(lib.actmi_op_conv3x3_c64(_p(x.contiguous()), _p(w16), float(w_scale), _p(scale.contiguous()), _p(bias.contiguous()),
                                     _p(res), _p(out), G, B, H, W, 1 if relu else 0, L.current_stream_ptr()), None,
            "op_conv3x3_c64")
    return out


def wgrad3x3_c64(dy, x, dy_scale=None):
    """dW [G][64][3][3][64] (O, kh, kw, I) of the 64 -> 64 channel 3x3 / s1 / p1 convolution from dy, x [G][B][H][W][64]."""
    G, B, H, W, Cc = x.shape
    assert Cc == 64 and dy.shape == x.shape
    nwg = max(1, min(256 // G, B * ((W + 31) // 32)))
    ws = torch.empty(G * nwg * 64 * 576, device=x.device, dtype=torch.float32)
    dw = torch.empty(G, 64, 3, 3, 64, device=x.device, dtype=torch.float32)
    lib = L.load()
    L.check(lib.actmi_op_wgrad3x3_c64(_p(dy.contiguous()), _p(x.contiguous()), _p(dw), _p(ws), ws.numel(),
                                      _p(dy_scale) if dy_scale is not None else None, G, B, H, W, L.current_stream_ptr()),
            None, "op_wgrad3x3_c64")
    return dw


def wgrad7x7s2(dy, x4, dy_scale=None):
    """dW [G][64][7][7][4] (O, kh, kw, I) of the stem convolution (7x7 / s2 / p3) from dy [G][B][Ho][Wo][64], x4 [G][B][H][W][4]."""
    G, B, H, W, Cc = x4.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    assert Cc == 4 and tuple(dy.shape) == (G, B, Ho, Wo, 64)
    nwg = max(1, min(512 // G, B * ((Wo + 31) // 32)))
    ws = torch.empty(G * nwg * 64 * 196, device=x4.device, dtype=torch.float32)
    dw = torch.empty(G, 64, 7, 7, 4, device=x4.device, dtype=torch.float32)
    lib = L.load()
    L.check(lib.actmi_op_wgrad7x7s2(_p(dy.contiguous()), _p(x4.contiguous()), _p(dw), _p(ws), ws.numel(),
                                    _p(dy_scale) if dy_scale is not None else None, G, B, H, W, L.current_stream_ptr()),
            None, "op_wgrad7x7s2")
    return dw



def attention(q, k, v, nheads, kpm=None, q_shared=False, want_lse=False, split=True, drop_p=0.0, drop_seed=0, prec=None,
              causal=False):
    """q [B,Nq,D] (or [Nq,D] when q_shared), k/v [B,Nk,D] (views with row stride allowed); returns [B,Nq,D]."""
    lib = L.load()
    B, Nk, D = k.shape
    Nq = q.shape[-2]
    hd = D // nheads
    out = torch.zeros((B, Nq, D), dtype=torch.float32, device=k.device)
    lse = torch.zeros((B, nheads, Nq), dtype=torch.float32, device=k.device) if want_lse else None
    d = L.AttnDesc()
    d.Q, d.q_bs, d.q_rs = q.data_ptr(), (0 if q_shared else q.stride(0)), q.stride(-2)
    d.K, d.k_bs, d.k_rs = k.data_ptr(), k.stride(0), k.stride(1)
    d.V, d.v_bs, d.v_rs = v.data_ptr(), v.stride(0), v.stride(1)
    d.O, d.o_bs, d.o_rs = out.data_ptr(), out.stride(0), out.stride(1)
    if kpm is not None:
        d.kpm, d.kpm_bs = kpm.data_ptr(), kpm.stride(0)
    d.lse = lse.data_ptr() if lse is not None else None
    if split:                                   # workspace for the split-KV path (used when the grid is small)
        ws = torch.empty(8 * B * Nq * (D + 2 * nheads), dtype=torch.float32, device=k.device)
        d.ws, d.ws_floats = ws.data_ptr(), ws.numel()
    d.B, d.H, d.Nq, d.Nk, d.HD = B, nheads, Nq, Nk, hd
    d.scale = 1.0 / (hd ** 0.5)
    d.drop_p, d.drop_seed = float(drop_p), int(drop_seed)
    d.prec = PREC[prec]
    d.causal = 1 if causal else 0
    L.check(lib.actmi_op_attention(C.byref(d), L.current_stream_ptr()), None, "op_attention")
    return (out, lse) if want_lse else out


def attention_bwd(q, k, v, out, lse, dout, nheads, kpm=None, drop_p=0.0, drop_seed=0, do_scale=None, want_amax=False):
    """dq, dk, dv of ``attention`` (f16x3, no materialised scores): q [B,Nq,D], k / v [B,Nk,D] (row-strided views allowed),
    out / dout [B,Nq,D] contiguous, lse [B,H,Nq] from ``attention(..., want_lse=True)``."""
    lib = L.load()
    B, Nk, D = k.shape
    Nq = q.shape[1]
    dq = torch.empty((B, Nq, D), dtype=torch.float32, device=k.device)
    dk = torch.empty((B, Nk, D), dtype=torch.float32, device=k.device)
    dv = torch.empty((B, Nk, D), dtype=torch.float32, device=k.device)
    ws = torch.empty(B * nheads * Nq, dtype=torch.float32, device=k.device)
    amax = torch.zeros(1, dtype=torch.int32, device=k.device) if want_amax else None
    d = L.AttnBwdDesc()
    d.q, d.q_bs, d.q_rs = q.data_ptr(), q.stride(0), q.stride(1)
    d.k, d.k_bs, d.k_rs = k.data_ptr(), k.stride(0), k.stride(1)
    d.v, d.v_bs, d.v_rs = v.data_ptr(), v.stride(0), v.stride(1)
    d.o, d.d_o, d.lse = out.data_ptr(), dout.data_ptr(), lse.data_ptr()
    d.dq, d.dq_bs, d.dq_rs = dq.data_ptr(), dq.stride(0), dq.stride(1)
    d.dk, d.dk_bs, d.dk_rs = dk.data_ptr(), dk.stride(0), dk.stride(1)
    d.dv, d.dv_bs, d.dv_rs = dv.data_ptr(), dv.stride(0), dv.stride(1)
    if kpm is not None:
        d.kpm, d.kpm_bs = kpm.data_ptr(), kpm.stride(0)
    d.B, d.H, d.Nq, d.Nk, d.HD = B, nheads, Nq, Nk, D // nheads
    d.drop_p, d.drop_seed = float(drop_p), int(drop_seed)
    d.delta_ws = ws.data_ptr()
    d.do_scale = do_scale.data_ptr() if do_scale is not None else None
    d.amax_out = amax.data_ptr() if amax is not None else None
    L.check(lib.actmi_op_attention_bwd(C.byref(d), L.current_stream_ptr()), None, "op_attention_bwd")
    return (dq, dk, dv, amax) if want_amax else (dq, dk, dv)


def layernorm(x, w, b, res=None, res_mod=0, w2=None, b2=None, eps=1e-5):
    lib = L.load()
    M, D = x.shape
    y = torch.empty_like(x)
    L.check(lib.actmi_op_layernorm(_p(x), _p(res), res_mod, _p(w), _p(b), _p(w2), _p(b2), _p(y), M, D, eps,
                                   L.current_stream_ptr()), None, "op_layernorm")
    return y


def maxpool3x3s2(x):
    """x [n,H,W,C] NHWC -> [n,Ho,Wo,C]."""
    lib = L.load()
    n, H, W, Cc = x.shape
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    y = torch.empty((n, Ho, Wo, Cc), dtype=torch.float32, device=x.device)
    L.check(lib.actmi_op_maxpool3x3s2(_p(x), _p(y), n, H, W, Cc, L.current_stream_ptr()), None, "op_maxpool")
    return y


def conv1(image, w_oihw, scale, bias, prec=None):
    """image u8 [B,C,H,W,3] or f32 [B,C,3,H,W]; w [C,Cout,3,7,7]; scale/bias [C,Cout] -> [C,B,Ho,Wo,Cout]."""
    lib = L.load()
    image = image.contiguous()
    if image.dtype == torch.uint8:
        B, Cn, H, W, _ = image.shape
        fmt = L.IMG_U8_NHWC
    else:
        B, Cn, _, H, W = image.shape
        fmt = L.IMG_F32_NCHW
    Cout = w_oihw.shape[1]
    Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    out = torch.zeros((Cn, B, Ho, Wo, Cout), dtype=torch.float32, device=image.device)
    ws = torch.empty(Cn * Cout * 148 + 768, dtype=torch.float32, device=image.device)
    L.check(lib.actmi_op_conv1(_p(image), fmt, _p(w_oihw), _p(scale), _p(bias), _p(out), _p(ws), B, Cn, H, W, Cout,
                               PREC[prec], L.current_stream_ptr()), None, "op_conv1")
    return out


# ---- DiffusionPolicy path (reference policy.py:20-241): the non-GEMM ops, channel-last ---------------------------------
ACT = {None: 0, "none": 0, "relu": 1, "mish": 2}


_GN_WS = {}


def groupnorm(x, weight, bias, groups, eps=1e-5, act=None, res=None, res_after=False, film=None, out=None):
    """x [n, P, C] (any leading spatial shape flattened into P is fine: pass [n, ..., C]); torch.nn.GroupNorm statistics.
    out = act(GN(x) + res) (res_after=False) or act(GN(x)) * film_scale + film_bias + res (res_after=True).
    `out`: optional contiguous tensor of x's shape to write into."""
    lib = L.load()
    x = x.contiguous()
    n, Cc = x.shape[0], x.shape[-1]
    P = x.numel() // (n * Cc)
    if out is None:
        out = torch.empty_like(x)
    assert out.is_contiguous() and out.shape == x.shape
    fs, fb = (film[0].contiguous(), film[1].contiguous()) if film is not None else (None, None)
    rm = 0 if res is None else (2 if res_after else 1)
    # chunk statistics of the large-map path (3 floats per sample, group, chunk): one workspace per (device, stream) -- two
    # streams normalising concurrently must not share it
    key = (x.device, torch.cuda.current_stream(x.device).cuda_stream)
    ws = _GN_WS.get(key)
    if ws is None:
        ws = _GN_WS[key] = torch.empty(1 << 20, dtype=torch.float32, device=x.device)
    L.check(lib.actmi_op_groupnorm(_p(x), _p(res.contiguous() if res is not None else None), _p(fs), _p(fb), _p(weight), _p(bias),
                                   _p(out), n, P, Cc, int(groups), float(eps), ACT[act], rm, _p(ws), ws.numel(),
                                   L.current_stream_ptr()), None, "op_groupnorm")
    return out


def spatial_softmax(logits, H, W, temperature=1.0):
    """logits [n, H*W, K] -> [n, K, 2] expected (x, y) keypoints (robomimic SpatialSoftmax)."""
    lib = L.load()
    logits = logits.contiguous()
    n, P, K = logits.shape
    assert P == H * W
    out = torch.empty((n, K, 2), dtype=torch.float32, device=logits.device)
    L.check(lib.actmi_op_spatial_softmax(_p(logits), _p(out), n, H, W, K, float(temperature), L.current_stream_ptr()), None,
            "op_spatial_softmax")
    return out


def unfold1d(x, k, stride=1, pad=0, transposed=False):
    """x [B, T, C] -> [B, To, k*C]: the rows a Conv1d (or, transposed, a ConvTranspose1d) contracts with its weights."""
    lib = L.load()
    x = x.contiguous()
    B, T, Cc = x.shape
    To = (T - 1) * stride - 2 * pad + k if transposed else (T + 2 * pad - k) // stride + 1
    out = torch.empty((B, To, k * Cc), dtype=torch.float32, device=x.device)
    L.check(lib.actmi_op_unfold1d(_p(x), _p(out), B, T, Cc, k, stride, pad, To, 1 if transposed else 0, L.current_stream_ptr()),
            None, "op_unfold1d")
    return out


def ddim_step(x, eps, alpha_t, alpha_prev, clip=True):
    """in place: diffusers DDIMScheduler.step with eta = 0, epsilon prediction."""
    lib = L.load()
    assert x.is_contiguous() and eps.is_contiguous() and x.numel() == eps.numel()
    L.check(lib.actmi_op_ddim_step(_p(x), _p(eps), x.numel(), float(alpha_t ** -0.5), float((1.0 - alpha_t) ** 0.5),
                                   float(alpha_prev ** 0.5), float(max(0.0, 1.0 - alpha_prev) ** 0.5), 1 if clip else 0,
                                   L.current_stream_ptr()), None, "op_ddim_step")
    return x


def mish(x):
    lib = L.load()
    x = x.contiguous()
    y = torch.empty_like(x)
    L.check(lib.actmi_op_mish(_p(x), _p(y), x.numel(), L.current_stream_ptr()), None, "op_mish")
    return y


def u8_to_nhwc4(image_u8):
    """u8 [B, Cam, H, W, 3] -> f32 [Cam, B, H, W, 4] in [0, 1] (channel 3 = 0): the GEMM convolution wants Cin % 4 == 0."""
    lib = L.load()
    image_u8 = image_u8.contiguous()
    B, Cam, H, W, _ = image_u8.shape
    out = torch.empty((Cam, B, H, W, 4), dtype=torch.float32, device=image_u8.device)
    L.check(lib.actmi_op_u8_to_nhwc4(_p(image_u8), _p(out), B, Cam, H, W, L.current_stream_ptr()), None, "op_u8_to_nhwc4")
    return out


# ---- pieces of the latent-prior training step (csrc/prior.hip; reference train_latent_model.py:323-343) -----------------
def gemm_t(A, Bm, ta=False, tb=False, bias=None, res=None, out=None, prec="f32"):
    """C = A' @ B'^T (+ bias + res) with the transposed operand forms of the backward: ta: A is stored [K][M]; tb: B is stored
    [K][N] (else [N][K], the nn.Linear layout).  Linear backward: dX = gemm_t(dY, W, tb=True); dW = gemm_t(dY, X, ta=True, tb=True)."""
    lib = L.load()
    if ta:
        K, M = A.shape
    else:
        M, K = A.shape
    N = Bm.shape[1] if tb else Bm.shape[0]
    assert (Bm.shape[0] if tb else Bm.shape[1]) == K, (A.shape, Bm.shape, ta, tb)
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=A.device)
    d = L.GemmDesc()
    d.A, d.lda, d.mode, d.ta = A.data_ptr(), A.stride(0), 0, 1 if ta else 0
    d.Bw, d.ldb, d.tb = Bm.data_ptr(), Bm.stride(0), 1 if tb else 0
    d.bias = bias.data_ptr() if bias is not None else None
    if res is not None:
        d.res, d.ldres = res.data_ptr(), res.stride(0)
    d.C, d.ldc = out.data_ptr(), out.stride(0)
    d.M, d.N, d.K, d.groups = M, N, K, 1
    d.prec = PREC[prec]
    L.check(lib.actmi_op_gemm(C.byref(d), L.current_stream_ptr()), None, "op_gemm")
    return out


def gelu(x):
    y = torch.empty_like(x)
    L.check(L.load().actmi_op_gelu(_p(x), _p(y), x.numel(), L.current_stream_ptr()), None, "op_gelu")
    return y


def gelu_bwd(x, dy):
    dx = torch.empty_like(x)
    L.check(L.load().actmi_op_gelu_bwd(_p(x), _p(dy), _p(dx), x.numel(), L.current_stream_ptr()), None, "op_gelu_bwd")
    return dx


def dropout(x, p, seed):
    """y = x * keep(seed, i) / (1 - p); the backward is the same call on the gradient"""
    y = torch.empty_like(x)
    L.check(L.load().actmi_op_dropout(_p(x), _p(y), x.numel(), float(p), int(seed), L.current_stream_ptr()), None, "op_dropout")
    return y


def small_attention(qkv, nheads, causal=True, drop_p=0.0, seed=0):
    """qkv [n,T,3D] (q | k | v) -> [n,T,D]; T <= 64, D / nheads <= 64"""
    n, T, D3 = qkv.shape
    D = D3 // 3
    out = torch.empty((n, T, D), dtype=torch.float32, device=qkv.device)
    L.check(L.load().actmi_op_small_attention(_p(qkv), _p(out), n, T, nheads, D // nheads, 1 if causal else 0, float(drop_p), int(seed),
                                              L.current_stream_ptr()), None, "op_small_attention")
    return out


def small_attention_bwd(qkv, dout, nheads, causal=True, drop_p=0.0, seed=0):
    n, T, D3 = qkv.shape
    D = D3 // 3
    dqkv = torch.empty_like(qkv)
    L.check(L.load().actmi_op_small_attention_bwd(_p(qkv), _p(dout), _p(dqkv), n, T, nheads, D // nheads, 1 if causal else 0,
                                                  float(drop_p), int(seed), L.current_stream_ptr()), None, "op_small_attention_bwd")
    return dqkv


def soft_ce_dim1(logits, target, want_grad=True):
    """F.cross_entropy(logits [B,T,V], target [B,T,V] probabilities): classes along dim 1.  -> (loss [1], dlogits or None)"""
    B, T, V = logits.shape
    loss = torch.empty(1, dtype=torch.float32, device=logits.device)
    ws = torch.empty(B * V, dtype=torch.float32, device=logits.device)
    dl = torch.empty_like(logits) if want_grad else None
    L.check(L.load().actmi_op_soft_ce_dim1(_p(logits), _p(target), B, T, V, _p(loss), _p(dl), _p(ws), L.current_stream_ptr()), None,
            "op_soft_ce_dim1")
    return loss, dl


def argmax_l1(logits, target):
    """mean |one_hot(argmax(logits, -1)) - target|"""
    V = logits.shape[-1]
    rows = logits.numel() // V
    out = torch.empty(1, dtype=torch.float32, device=logits.device)
    ws = torch.empty(rows, dtype=torch.float32, device=logits.device)
    L.check(L.load().actmi_op_argmax_l1(_p(logits), _p(target), rows, V, _p(out), _p(ws), L.current_stream_ptr()), None, "op_argmax_l1")
    return out


def layernorm_bwd(x, w, dy, dw, db, ws, dx_add=None, eps=1e-5):
    """-> dx; dw += , db += (views of the gradient arena)"""
    M, D = x.shape
    dx = torch.empty_like(x)
    L.check(L.load().actmi_op_layernorm_bwd(_p(x), _p(w), _p(dy), _p(dx_add), _p(dx), _p(dw), _p(db), M, D, eps, _p(ws), ws.numel(),
                                            L.current_stream_ptr()), None, "op_layernorm_bwd")
    return dx


def colsum(src, out, ws):
    """out[n] += sum_m src[m][n]"""
    M, N = src.shape
    L.check(L.load().actmi_op_colsum(_p(src), src.stride(0), _p(out), M, N, _p(ws), ws.numel(), L.current_stream_ptr()), None, "op_colsum")


def sum_batch(src, dst, accumulate=False):
    """src [B,R,D] -> dst[R,D] (+)= sum_b src[b]"""
    B, Rr, D = src.shape
    L.check(L.load().actmi_op_sum_batch(_p(src), src.stride(0), src.stride(1), _p(dst), B, Rr, D, 1 if accumulate else 0,
                                        L.current_stream_ptr()), None, "op_sum_batch")


def adamw(p, g, m, v, lr, weight_decay, step, betas=(0.9, 0.999), eps=1e-8):
    L.check(L.load().actmi_op_adamw(_p(p), _p(g), _p(m), _p(v), p.numel(), float(lr), float(weight_decay), float(betas[0]),
                                    float(betas[1]), float(eps), int(step), L.current_stream_ptr()), None, "op_adamw")


class TemporalEnsemble:
    """Batched temporal ensembling state for E episodes (reference imitate_episodes.py:338-339, 402-411).
    Ring buffer [E,Q,Q,A] instead of the reference's [T,T+Q,A] per episode: only the last Q chunks can
    contribute to step t."""

    def __init__(self, num_episodes, num_queries, action_dim=16, k=0.01, device="cuda:0"):
        self.E, self.Q, self.A, self.k = num_episodes, num_queries, action_dim, float(k)
        self.ring = torch.zeros((self.E, self.Q, self.Q, self.A), dtype=torch.float32, device=device)
        self.t = torch.zeros((self.E,), dtype=torch.int32, device=device)
        self.out = torch.zeros((self.E, self.A), dtype=torch.float64, device=device)
        self.populated = torch.zeros((self.E, self.Q), dtype=torch.uint8, device=device)

    def reset(self):
        self.ring.zero_()
        self.t.zero_()

    def step(self, all_actions):
        """all_actions [E,Q,A] f32 cuda -> raw_action [E,A] f64 (same dtype as the reference's raw_action)."""
        lib = L.load()
        a = all_actions.contiguous()
        dev = self.ring.device
        if a.device != dev:
            raise ValueError(f"all_actions lives on {a.device} but this ensemble is bound to {dev}")
        with torch.cuda.device(dev):          # the kernel launches on the ring's device whatever the caller left current
            L.check(lib.actmi_ensemble_step(_p(self.ring), _p(self.t), _p(a), self.k, _p(self.out), _p(self.populated),
                                            self.E, self.Q, self.A, L.current_stream_ptr()), None, "ensemble_step")
        return self.out
