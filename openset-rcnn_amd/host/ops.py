"""Thin torch-tensor wrappers over the C ABI (include/osr.h). torch is plumbing here: device memory, the
current HIP stream and output allocation. Every op runs on the HIP library or raises; there is no eager path."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import ConvParams, OsrError, Pyramid, RpnLevels, check

_DT = {torch.float32: _lib.OSR_F32, torch.float16: _lib.OSR_F16, torch.bfloat16: _lib.OSR_BF16}


def _p(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need(t: torch.Tensor, dtype=None, name="tensor"):
    if not t.is_cuda:
        raise OsrError(f"{name} must live on the GPU (got {t.device}); the HIP path has no CPU fallback")
    if dtype is not None and t.dtype != dtype:
        raise OsrError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise OsrError(f"{name} must be contiguous")
    return t


def dtype_code(dt: torch.dtype) -> int:
    return _DT[dt]


# ----------------------------------------------------------------------------------------------------------
def stem_padded_width(wp: int) -> int:
    return int(_lib.load().osr_stem_padded_width(wp))


def preprocess(images: torch.Tensor, hp: int, wp: int, mean, std, dtype=torch.float16) -> torch.Tensor:
    """(n,3,h,w) uint8/float32 -> normalised, zero-haloed (n, hp+6, wpad, 4) NHWC tensor for the stem."""
    lib = _lib.load()
    _need(images, name="images")
    if images.dtype not in (torch.uint8, torch.float32):
        raise OsrError("images must be uint8 or float32")
    n, c, h, w = images.shape
    assert c == 3
    out = torch.empty((n, hp + 6, stem_padded_width(wp), 4), dtype=dtype, device=images.device)
    m = (C.c_float * 3)(*[float(v) for v in mean])
    s = (C.c_float * 3)(*[float(v) for v in std])
    check(lib.osr_preprocess(_p(images), int(images.dtype == torch.uint8), n, h, w, hp, wp, m, s, _p(out), _DT[dtype], _stream()),
          "osr_preprocess")
    return out


_CONCURRENCY = [1]
# bench.py's train_step leg sets this to a dict {"conv": 0.0, "wgrad": 0.0}: algorithmic FLOPs (2*M*K*N of the layer definition)
# of every MFMA conv / FC launch (forward and data-gradient launches both go through conv2d) and of every weight-gradient launch
FLOP_COUNT = None
# osr_conv2d_fwd may cut the partial last dispatch round of a deep-K 1x1 / FC layer along K when it is handed a workspace
# (include/osr.h); False keeps every layer a single launch (A/B runs)
SPLIT_K_TAIL = True


class concurrent_streams:
    """Context manager: the launches inside run on `n` HIP streams side by side (the engine's micro-batch streams). Passed to
    the conv kernels as osr_conv_params.concurrency, a tile-selection hint (results do not depend on it)."""

    def __init__(self, n: int):
        self.n = int(n)

    def __enter__(self):
        self.prev = _CONCURRENCY[0]
        _CONCURRENCY[0] = self.n
        return self

    def __exit__(self, *exc):
        _CONCURRENCY[0] = self.prev
        return False


def _new_conv_params() -> "ConvParams":
    p = ConvParams()
    p.concurrency = _CONCURRENCY[0]
    return p


def conv2d(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, stride: int = 1, pad: int = 0, relu: bool = False,
           residual: Optional[torch.Tensor] = None, res_mode: int = 0, out_dtype: Optional[torch.dtype] = None,
           out: Optional[torch.Tensor] = None, post_mask: Optional[torch.Tensor] = None,
           row_seg: Optional[Tuple[torch.Tensor, int]] = None) -> torch.Tensor:
    """NHWC implicit-GEMM convolution. x (n,h,w,cin) f16/bf16; weight (cout,kh,kw,cin) same dtype; bias f32.
    post_mask (n,ho,wo,cout), x's dtype: the result is zeroed where post_mask <= 0, in the same launch
    (osr_conv2d_fwd_masked; needs cin % 64 == 0, else the mask is applied by a second launch).
    row_seg = (counts int32 (s,), rows per segment): the output rows are s segments of which only the first counts[i] rows carry
    data (padded per-image lists); tiles without a data row are skipped and their rows left unwritten."""
    lib = _lib.load()
    _need(x, name="x"); _need(weight, x.dtype, "weight"); _need(bias, torch.float32, "bias")
    n, hi, wi, cin = x.shape
    cout, kh, kw, cin2 = weight.shape
    if cin2 != cin:
        raise OsrError(f"weight cin {cin2} != input cin {cin}")
    ho = (hi + 2 * pad - kh) // stride + 1
    wo = (wi + 2 * pad - kw) // stride + 1
    out_dtype = out_dtype or x.dtype
    if out is None:
        out = torch.empty((n, ho, wo, cout), dtype=out_dtype, device=x.device)
    else:
        _need(out, out_dtype, "out")
        assert out.numel() == n * ho * wo * cout
    if FLOP_COUNT is not None:
        FLOP_COUNT["conv"] += 2.0 * n * ho * wo * cout * kh * kw * cin
    p = _new_conv_params()
    p.n, p.hi, p.wi, p.cin, p.ho, p.wo, p.cout = n, hi, wi, cin, ho, wo, cout
    p.kh, p.kw, p.stride_h, p.stride_w, p.pad_h, p.pad_w = kh, kw, stride, stride, pad, pad
    p.in_stride_n, p.in_stride_h, p.in_stride_w = hi * wi * cin, wi * cin, cin
    p.out_stride_n, p.out_stride_h, p.out_stride_w = ho * wo * cout, wo * cout, cout
    p.relu, p.res_mode, p.pad_mode = int(relu), int(res_mode), 0
    p.in_dtype, p.out_dtype = _DT[x.dtype], _DT[out_dtype]
    if row_seg is not None:
        counts, seg_rows = row_seg
        _need(counts, torch.int32, "row_seg counts")
        if seg_rows < 1 or counts.numel() * seg_rows < n * ho * wo:
            raise OsrError("row_seg does not cover the output rows")
        p.row_seg_counts, p.row_seg_rows = counts.data_ptr(), int(seg_rows)
    if res_mode:
        _need(residual, x.dtype, "residual")
        rn, rh, rw, rc = residual.shape
        exp = (n, ho, wo, cout) if res_mode in (1, 3) else (n, (ho + 1) // 2, (wo + 1) // 2, cout)
        if (rn, rh, rw, rc) != exp:
            raise OsrError(f"residual shape {tuple(residual.shape)} != expected {exp} for res_mode {res_mode}")
        p.res_stride_n, p.res_stride_h, p.res_stride_w = rh * rw * rc, rw * rc, rc
    if post_mask is not None:
        _need(post_mask, x.dtype, "post_mask")
        if tuple(post_mask.shape) != (n, ho, wo, cout):
            raise OsrError(f"post_mask shape {tuple(post_mask.shape)} != {(n, ho, wo, cout)}")
        st = lib.osr_conv2d_fwd_masked(C.byref(p), _p(x), _p(weight), _p(bias), _p(residual) if res_mode else None, _p(post_mask), _p(out),
                                       _stream())
        if st != _lib.ERR_UNSUPPORTED:
            check(st, "osr_conv2d_fwd_masked")
            return out
    ws = None
    if SPLIT_K_TAIL and post_mask is None and res_mode == 0 and stride == 1:
        wsb = int(lib.osr_conv2d_fwd_workspace_bytes(C.byref(p)))  # > 0: a deep-K 1x1 / FC layer whose last dispatch round is mostly empty
        if wsb > 0:
            ws = torch.empty((wsb,), dtype=torch.uint8, device=x.device)
            p.workspace, p.workspace_bytes = ws.data_ptr(), wsb
    check(lib.osr_conv2d_fwd(C.byref(p), _p(x), _p(weight), _p(bias), _p(residual) if res_mode else None, _p(out), _stream()),
          "osr_conv2d_fwd")
    if post_mask is not None:
        relu_mask_(out, post_mask)
    return out


def stem_conv(xpad: torch.Tensor, w_view: torch.Tensor, bias: torch.Tensor, hp: int, wp: int, relu: bool = True) -> torch.Tensor:
    """7x7/s2/p3 stem on the pre-padded NHWC4 image (see preprocess): a (kh=7, kw=1, cin=32) implicit GEMM whose
    'pixels' are 4-element groups, 8 of which (7 taps + a zero-weight one) form one K slice. With the 8-row weight
    view (pack_stem_weight) K = 256 and the BK=64 kernel takes it; the 8th row reads the halo with zero weights."""
    lib = _lib.load()
    _need(xpad, name="xpad"); _need(w_view, xpad.dtype, "w_view"); _need(bias, torch.float32, "bias")
    n, hd, wd, c4 = xpad.shape
    assert c4 == 4 and hd == hp + 6 and wd == stem_padded_width(wp)
    cout = w_view.shape[0]
    assert tuple(w_view.shape[1:]) in ((7, 1, 32), (8, 1, 32))
    kh = w_view.shape[1]
    ho, wo = hp // 2, wp // 2
    out = torch.empty((n, ho, wo, cout), dtype=xpad.dtype, device=xpad.device)
    if FLOP_COUNT is not None:
        FLOP_COUNT["conv"] += 2.0 * n * ho * wo * cout * 147  # 7*7*3 real taps
    p = _new_conv_params()
    p.n, p.hi, p.wi, p.cin, p.ho, p.wo, p.cout = n, hd, wd, 32, ho, wo, cout
    p.kh, p.kw, p.stride_h, p.stride_w, p.pad_h, p.pad_w = kh, 1, 2, 2, 0, 0
    p.in_stride_n, p.in_stride_h, p.in_stride_w = hd * wd * 4, wd * 4, 4
    p.out_stride_n, p.out_stride_h, p.out_stride_w = ho * wo * cout, wo * cout, cout
    p.relu, p.res_mode, p.pad_mode = int(relu), 0, 1
    p.in_dtype = p.out_dtype = _DT[xpad.dtype]
    check(lib.osr_conv2d_fwd(C.byref(p), _p(xpad), _p(w_view), _p(bias), None, _p(out), _stream()), "osr_conv2d_fwd(stem)")
    return out


def stem_maxpool(xpad: torch.Tensor, w_view: torch.Tensor, bias: torch.Tensor, hp: int, wp: int) -> torch.Tensor:
    """[d2] BasicStem in one launch (osr_stem_maxpool_fwd): 7x7/s2/p3 conv + ReLU + 3x3/s2/p1 max pool on the pre-padded NHWC4 image;
    the (n, hp/2, wp/2, 64) stem output stays in LDS. Same bits as stem_conv + maxpool3x3s2. f16 / bf16."""
    lib = _lib.load()
    _need(xpad, name="xpad"); _need(w_view, xpad.dtype, "w_view"); _need(bias, torch.float32, "bias")
    n, hd, wd, c4 = xpad.shape
    assert c4 == 4 and hd == hp + 6 and wd == stem_padded_width(wp) and hp % 2 == 0 and wp % 2 == 0
    assert w_view.shape[0] == 64 and tuple(w_view.shape[2:]) == (1, 32) and w_view.shape[1] in (7, 8)
    hs, ws = hp // 2, wp // 2
    out = torch.empty((n, (hs - 1) // 2 + 1, (ws - 1) // 2 + 1, 64), dtype=xpad.dtype, device=xpad.device)
    if FLOP_COUNT is not None:
        FLOP_COUNT["conv"] += 2.0 * n * hs * ws * 64 * 147  # 7*7*3 real taps of every stem pixel (the pool's halo recompute is not credited)
    check(lib.osr_stem_maxpool_fwd(_p(xpad), n, hp, wp, _p(w_view), int(w_view.shape[1]), _p(bias), _p(out), _DT[xpad.dtype], _stream()), "osr_stem_maxpool_fwd")
    return out


def stem_maxpool_raw(images: torch.Tensor, hp: int, wp: int, mean, std, w_view: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """preprocess + stem_maxpool in one launch (osr_stem_maxpool_fwd_raw): (n,3,h,w) uint8 / float32 images -> (n, hp/4, wp/4, 64) in
    w_view's dtype; the normalised, padded copy of the batch is never written. Same bits as preprocess -> stem_maxpool."""
    lib = _lib.load()
    _need(images, name="images"); _need(w_view, name="w_view"); _need(bias, torch.float32, "bias")
    if images.dtype not in (torch.uint8, torch.float32) or w_view.dtype not in (torch.float16, torch.bfloat16):
        raise OsrError("stem_maxpool_raw: uint8 / float32 images, f16 / bf16 weights")
    n, c, h, w = images.shape
    assert c == 3 and hp >= h and wp >= w and hp % 2 == 0 and wp % 2 == 0
    assert w_view.shape[0] == 64 and tuple(w_view.shape[2:]) == (1, 32) and w_view.shape[1] in (7, 8)
    hs, ws = hp // 2, wp // 2
    out = torch.empty((n, (hs - 1) // 2 + 1, (ws - 1) // 2 + 1, 64), dtype=w_view.dtype, device=images.device)
    if FLOP_COUNT is not None:
        FLOP_COUNT["conv"] += 2.0 * n * hs * ws * 64 * 147
    m = (C.c_float * 3)(*[float(v) for v in mean])
    s = (C.c_float * 3)(*[float(v) for v in std])
    check(lib.osr_stem_maxpool_fwd_raw(_p(images), int(images.dtype == torch.uint8), n, h, w, hp, wp, m, s, _p(w_view), int(w_view.shape[1]), _p(bias),
                                       _p(out), _DT[w_view.dtype], _stream()), "osr_stem_maxpool_fwd_raw")
    return out


def linear(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, relu: bool = False,
           out_dtype: Optional[torch.dtype] = None, row_seg: Optional[Tuple[torch.Tensor, int]] = None) -> torch.Tensor:
    """Fully connected layer on the MFMA path: x (m,k) f16/bf16, weight (n,k). row_seg: see conv2d."""
    m, k = x.shape
    y = conv2d(x.view(1, m, 1, k), weight.view(weight.shape[0], 1, 1, k), bias, relu=relu, out_dtype=out_dtype, row_seg=row_seg)
    return y.view(m, weight.shape[0])


def conv2d_chain(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, w3: torch.Tensor, b3: torch.Tensor, residual: torch.Tensor,
                 stride: int = 1, pad: int = 0, keep_mid: bool = False):
    """A bottleneck's conv2 -> conv3 in one launch (osr_conv2d_chain_fwd): relu(conv1x1(relu(conv(x, weight) + bias), w3) + b3 +
    residual). x (n,h,w,cin) f16/bf16, weight (cmid,kh,kw,cin), w3 (cout3,1,1,cmid), residual (n,ho,wo,cout3). Returns None when the
    shape is outside the fused kernel's envelope (the caller then runs conv2d twice). keep_mid: also return the first convolution's
    activated output (n,ho,wo,cmid) -- (out, mid) -- as the training step needs it."""
    lib = _lib.load()
    _need(x, name="x")
    if x.dtype not in (torch.float16, torch.bfloat16):
        return None
    _need(weight, x.dtype, "weight"); _need(w3, x.dtype, "w3"); _need(residual, x.dtype, "residual")
    _need(bias, torch.float32, "bias"); _need(b3, torch.float32, "b3")
    n, hi, wi, cin = x.shape
    cmid, kh, kw, cin2 = weight.shape
    cout3 = w3.shape[0]
    if cin2 != cin or tuple(w3.shape) != (cout3, 1, 1, cmid) or bias.numel() != cmid or b3.numel() != cout3:
        raise OsrError("conv2d_chain: weight shapes do not form conv -> 1x1 conv")
    ho = (hi + 2 * pad - kh) // stride + 1
    wo = (wi + 2 * pad - kw) // stride + 1
    if tuple(residual.shape) != (n, ho, wo, cout3):
        raise OsrError(f"conv2d_chain: residual shape {tuple(residual.shape)} != {(n, ho, wo, cout3)}")
    p = _new_conv_params()
    p.n, p.hi, p.wi, p.cin, p.ho, p.wo, p.cout = n, hi, wi, cin, ho, wo, cmid
    p.kh, p.kw, p.stride_h, p.stride_w, p.pad_h, p.pad_w = kh, kw, stride, stride, pad, pad
    p.in_stride_n, p.in_stride_h, p.in_stride_w = hi * wi * cin, wi * cin, cin
    p.out_stride_n, p.out_stride_h, p.out_stride_w = ho * wo * cmid, wo * cmid, cmid
    p.relu, p.res_mode, p.pad_mode = 1, 0, 0
    p.in_dtype = p.out_dtype = _DT[x.dtype]
    out = torch.empty((n, ho, wo, cout3), dtype=x.dtype, device=x.device)
    mid = torch.empty((n, ho, wo, cmid), dtype=x.dtype, device=x.device) if keep_mid else None
    st = lib.osr_conv2d_chain_fwd_ex(C.byref(p), _p(x), _p(weight), _p(bias), _p(w3), _p(b3), cout3, _p(residual), _p(out), _p(mid), _stream())
    if st == _lib.ERR_UNSUPPORTED:
        return None
    check(st, "osr_conv2d_chain_fwd")
    if FLOP_COUNT is not None:
        FLOP_COUNT["conv"] += 2.0 * n * ho * wo * (cmid * kh * kw * cin + cout3 * cmid)
    return (out, mid) if keep_mid else out


def bottleneck(x: torch.Tensor, w1, b1, w2, b2, w3, b3, wsc=None, bsc=None) -> Optional[torch.Tensor]:
    """One whole bottleneck block in one launch (osr_bottleneck_fwd): x (n,h,w,cin) f16/bf16, weights packed (cout,kh,kw,cin) in
    x's dtype, biases fp32; wsc / bsc = the projection shortcut (None: identity). Returns y (n,h,w,cout), or None when the shape
    is outside the fused kernel's envelope (the caller then runs the separate convolutions)."""
    lib = _lib.load()
    _need(x, name="x")
    if x.dtype not in (torch.float16, torch.bfloat16):
        return None
    n, h, w, cin = x.shape
    cmid, cout = w1.shape[0], w3.shape[0]
    for t, nm in ((w1, "w1"), (w2, "w2"), (w3, "w3")):
        _need(t, x.dtype, nm)
    for t, nm in ((b1, "b1"), (b2, "b2"), (b3, "b3")):
        _need(t, torch.float32, nm)
    if tuple(w1.shape) != (cmid, 1, 1, cin) or tuple(w2.shape) != (cmid, 3, 3, cmid) or tuple(w3.shape) != (cout, 1, 1, cmid):
        raise OsrError("bottleneck: weight shapes do not form a 1x1 -> 3x3 -> 1x1 block")
    p = _lib.BottleneckParams()
    p.n, p.h, p.w, p.cin, p.cmid, p.cout = n, h, w, cin, cmid, cout
    p.dtype, p.has_proj = _DT[x.dtype], int(wsc is not None)
    if wsc is not None:
        _need(wsc, x.dtype, "wsc"); _need(bsc, torch.float32, "bsc")
        if tuple(wsc.shape) != (cout, 1, 1, cin):
            raise OsrError("bottleneck: projection weight shape")
    out = torch.empty((n, h, w, cout), dtype=x.dtype, device=x.device)
    st = lib.osr_bottleneck_fwd(C.byref(p), _p(x), _p(w1), _p(b1), _p(w2), _p(b2), _p(w3), _p(b3), _p(wsc), _p(bsc), _p(out), _stream())
    if st == _lib.ERR_UNSUPPORTED:
        return None
    check(st, "osr_bottleneck_fwd")
    if FLOP_COUNT is not None:
        FLOP_COUNT["conv"] += 2.0 * n * h * w * (cin * cmid + 9 * cmid * cmid + cmid * cout + (cin * cout if wsc is not None else 0))
    return out


def resize_bilinear_u8(img: torch.Tensor, xbounds, xcoef, kx: int, ybounds, ycoef, ky: int, y_first: int, y_rows: int, nh: int, nw: int,
                       out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """PIL Image.resize(BILINEAR) of an (h, w, 3) uint8 CUDA image with Pillow's coefficient tables (host/data.py:
    pil_resample_coeffs) -> (3, nh, nw) uint8 (osr_resize_bilinear_u8)."""
    lib = _lib.load()
    _need(img, torch.uint8, "img")
    for t, nm in ((xbounds, "xbounds"), (xcoef, "xcoef"), (ybounds, "ybounds"), (ycoef, "ycoef")):
        _need(t, torch.int32, nm)
    h, w, c = img.shape
    if c != 3 or xbounds.shape[0] != nw or ybounds.shape[0] != nh or xcoef.shape[1] != kx or ycoef.shape[1] != ky:
        raise OsrError("resize_bilinear_u8: table shapes do not match the sizes")
    tmp_bytes = int(lib.osr_resize_tmp_bytes(y_rows, nw))
    tmp = torch.empty((tmp_bytes,), dtype=torch.uint8, device=img.device)
    if out is None:
        out = torch.empty((3, nh, nw), dtype=torch.uint8, device=img.device)
    else:
        _need(out, torch.uint8, "out")
        if tuple(out.shape) != (3, nh, nw):
            raise OsrError("resize_bilinear_u8: out must be (3, nh, nw)")
    check(lib.osr_resize_bilinear_u8(_p(img), h, w, w * 3, _p(xbounds), _p(xcoef), kx, _p(ybounds), _p(ycoef), ky, y_first, y_rows, nh, nw, _p(tmp),
                                     tmp_bytes, _p(out), _stream()), "osr_resize_bilinear_u8")
    return out


def maxpool3x3s2(x: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    _need(x, name="x")
    n, h, w, c = x.shape
    out = torch.empty((n, (h - 1) // 2 + 1, (w - 1) // 2 + 1, c), dtype=x.dtype, device=x.device)
    check(lib.osr_maxpool3x3s2(_p(x), n, h, w, c, _p(out), _DT[x.dtype], _stream()), "osr_maxpool3x3s2")
    return out


def subsample2(x: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    _need(x, name="x")
    n, h, w, c = x.shape
    out = torch.empty((n, (h - 1) // 2 + 1, (w - 1) // 2 + 1, c), dtype=x.dtype, device=x.device)
    check(lib.osr_subsample2(_p(x), n, h, w, c, _p(out), _DT[x.dtype], _stream()), "osr_subsample2")
    return out


def gemm_f32(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], relu: bool = False) -> torch.Tensor:
    lib = _lib.load()
    _need(a, torch.float32, "a"); _need(w, torch.float32, "w")
    if bias is not None:
        _need(bias, torch.float32, "bias")
    m, k = a.shape
    n = w.shape[0]
    assert w.shape[1] == k
    out = torch.empty((m, n), dtype=torch.float32, device=a.device)
    check(lib.osr_gemm_f32(_p(a), k, _p(w), _p(bias), _p(out), n, m, n, k, int(relu), _stream()), "osr_gemm_f32")
    return out


def gemm_f32_tn(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[m, n] = sum_k a[k, m] * b[k, n] (= a^T b) in exact fp32: the weight gradient dy^T x of an fp32 linear layer."""
    lib = _lib.load()
    _need(a, torch.float32, "a"); _need(b, torch.float32, "b")
    k, m = a.shape
    assert b.shape[0] == k
    n = b.shape[1]
    if out is None:
        out = torch.empty((m, n), dtype=torch.float32, device=a.device)
    else:
        _need(out, torch.float32, "out")
        assert tuple(out.shape) == (m, n)
    nb = int(lib.osr_gemm_f32_tn_workspace_bytes(m, n, k))
    ws = torch.empty((nb,), dtype=torch.uint8, device=a.device) if nb else None
    check(lib.osr_gemm_f32_tn(_p(a), m, _p(b), n, _p(out), n, m, n, k, _p(ws), nb, _stream()), "osr_gemm_f32_tn")
    return out


def cfrpn_head_tail(t: torch.Tensor, w_delta, b_delta, w_ctr, b_ctr) -> Tuple[torch.Tensor, torch.Tensor]:
    lib = _lib.load()
    _need(t, name="t")
    rows, c = t.shape
    for x in (w_delta, b_delta, w_ctr, b_ctr):
        _need(x, torch.float32, "head weight")
    deltas = torch.empty((rows, 4), dtype=torch.float32, device=t.device)
    ctr = torch.empty((rows,), dtype=torch.float32, device=t.device)
    check(lib.osr_cfrpn_head_tail(_p(t), _DT[t.dtype], rows, c, _p(w_delta), _p(b_delta), _p(w_ctr), _p(b_ctr), _p(deltas), _p(ctr),
                                  _stream()), "osr_cfrpn_head_tail")
    return deltas, ctr


def cfrpn_head_fused(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, w_tail: torch.Tensor, b_tail: torch.Tensor,
                     deltas_out: Optional[torch.Tensor] = None, ctr_out: Optional[torch.Tensor] = None, hidden_out: Optional[torch.Tensor] = None):
    """ClsFreeRPNHead.forward for one level in one launch. x (n,h,w,256) f16/bf16, weight (256,3,3,256) packed,
    w_tail (5,256) fp32 [deltas rows 0-3, centerness row 4], b_tail (5). Returns deltas (n*h*w, 4), ctr (n*h*w).
    hidden_out (n*h*w, 256), x's dtype: also receives the hidden state relu(conv + bias) (the training step keeps it)."""
    lib = _lib.load()
    _need(x, name="x"); _need(weight, x.dtype, "weight"); _need(bias, torch.float32, "bias")
    _need(w_tail, torch.float32, "w_tail"); _need(b_tail, torch.float32, "b_tail")
    n, hi, wi, cin = x.shape
    cout, kh, kw, _ = weight.shape
    pad = kh // 2
    rows = n * hi * wi
    if FLOP_COUNT is not None:
        FLOP_COUNT["conv"] += 2.0 * rows * cout * (kh * kw * cin + 5)  # the 3x3 conv + the five 1x1 outputs of the fused tail
    deltas = deltas_out if deltas_out is not None else torch.empty((rows, 4), dtype=torch.float32, device=x.device)
    ctr = ctr_out if ctr_out is not None else torch.empty((rows,), dtype=torch.float32, device=x.device)
    _need(deltas, torch.float32, "deltas"); _need(ctr, torch.float32, "ctr")
    assert deltas.numel() == rows * 4 and ctr.numel() == rows
    p = _new_conv_params()
    p.n, p.hi, p.wi, p.cin, p.ho, p.wo, p.cout = n, hi, wi, cin, hi, wi, cout
    p.kh, p.kw, p.stride_h, p.stride_w, p.pad_h, p.pad_w = kh, kw, 1, 1, pad, pad
    p.in_stride_n, p.in_stride_h, p.in_stride_w = hi * wi * cin, wi * cin, cin
    p.relu, p.res_mode, p.pad_mode = 1, 0, 0
    p.in_dtype = p.out_dtype = _DT[x.dtype]
    if hidden_out is not None:
        _need(hidden_out, x.dtype, "hidden_out")
        assert hidden_out.numel() == rows * 256
    check(lib.osr_cfrpn_head_fwd_ex(C.byref(p), _p(x), _p(weight), _p(bias), _p(w_tail), _p(b_tail), _p(deltas), _p(ctr), _p(hidden_out), _stream()),
          "osr_cfrpn_head_fwd")
    return deltas, ctr


def conv2d_pair(x: torch.Tensor, w_a: torch.Tensor, b_a: torch.Tensor, relu_a: bool, w_b: torch.Tensor, b_b: torch.Tensor, relu_b: bool,
                stride: int = 1, pad: int = 0):
    """Two convolutions of the same input and geometry in ONE launch (osr_conv2d_fwd_pair: the projection shortcut and conv1 of a
    stage's first bottleneck). Returns (out_a, out_b), or None outside the launch's envelope (the caller runs conv2d twice)."""
    lib = _lib.load()
    ca, kh, kw, cin = w_a.shape
    cb = w_b.shape[0]
    if tuple(w_b.shape[1:]) != (kh, kw, cin) or ca % 128 or cb % 128 or cin % 64 or x.dtype not in (torch.float16, torch.bfloat16):
        return None
    _need(x, name="x"); _need(w_a, x.dtype, "w_a"); _need(w_b, x.dtype, "w_b"); _need(b_a, torch.float32, "b_a"); _need(b_b, torch.float32, "b_b")
    n, hi, wi, _ = x.shape
    ho, wo = (hi + 2 * pad - kh) // stride + 1, (wi + 2 * pad - kw) // stride + 1
    out_a = torch.empty((n, ho, wo, ca), dtype=x.dtype, device=x.device)
    out_b = torch.empty((n, ho, wo, cb), dtype=x.dtype, device=x.device)
    p = _conv_params(n, hi, wi, cin, ho, wo, ca + cb, kh, kw, stride, pad, x.dtype, x.dtype)
    st = lib.osr_conv2d_fwd_pair(C.byref(p), _p(x), _p(w_a), _p(b_a), ca, int(relu_a), _p(out_a), _p(w_b), _p(b_b), cb, int(relu_b), _p(out_b), _stream())
    if st == _lib.ERR_UNSUPPORTED:
        return None
    check(st, "osr_conv2d_fwd_pair")
    if FLOP_COUNT is not None:
        FLOP_COUNT["conv"] += 2.0 * n * ho * wo * (ca + cb) * kh * kw * cin
    return out_a, out_b


def _conv_levels(xs: Sequence[torch.Tensor], outs, deltas=None, ctrs=None, weights=None, biases=None):
    assert 1 <= len(xs) <= _lib.MAX_CONV_LEVELS
    arr = (_lib.ConvLevel * len(xs))()
    for i, x in enumerate(xs):
        n, hi, wi, _ = x.shape
        arr[i].in_ = x.data_ptr()
        arr[i].out = outs[i].data_ptr() if outs is not None and outs[i] is not None else None
        arr[i].deltas = deltas[i].data_ptr() if deltas is not None else None
        arr[i].ctr = ctrs[i].data_ptr() if ctrs is not None else None
        arr[i].weight = weights[i].data_ptr() if weights is not None else None
        arr[i].bias = biases[i].data_ptr() if biases is not None else None
        arr[i].n, arr[i].hi, arr[i].wi = n, hi, wi
    return arr


def _levels_params(x0: torch.Tensor, weight: torch.Tensor, relu: bool):
    cout, kh, kw, cin = weight.shape
    assert kh == kw and kh % 2 == 1 and x0.shape[3] == cin
    p = _new_conv_params()
    p.cin, p.cout, p.kh, p.kw, p.stride_h, p.stride_w, p.pad_h, p.pad_w = cin, cout, kh, kw, 1, 1, kh // 2, kw // 2
    p.in_stride_w, p.out_stride_w = cin, cout
    p.relu, p.res_mode, p.pad_mode = 1 if relu else 0, 0, 0
    p.in_dtype = p.out_dtype = _DT[x0.dtype]
    return p


def conv2d_levels(xs: Sequence[torch.Tensor], weight, bias, relu: bool = False, outs: Optional[Sequence[torch.Tensor]] = None):
    """Stride-1, same-padding convolutions of one shape on several NHWC feature maps in ONE launch (osr_conv2d_fwd_levels: the FPN's output
    convs). weight / bias: one tensor shared by all levels, or a list with one per level. Returns the list of outputs, or None when the
    shape is outside the launch's envelope (cout % 256, cin % 64, f16 / bf16): the caller then runs conv2d per level."""
    lib = _lib.load()
    ws = list(weight) if isinstance(weight, (list, tuple)) else None
    bs = list(bias) if isinstance(bias, (list, tuple)) else None
    w0 = ws[0] if ws is not None else weight
    cout, kh, kw, cin = w0.shape
    if cout % 256 or cin % 64 or kh != kw or kh % 2 == 0 or kh * kw * cin // 64 < 2 or xs[0].dtype not in (torch.float16, torch.bfloat16) or len(xs) > _lib.MAX_CONV_LEVELS:
        return None
    assert (ws is None) == (bs is None) and (ws is None or (len(ws) == len(xs) and len(bs) == len(xs)))
    for w_, b_ in zip(ws if ws is not None else [weight], bs if bs is not None else [bias]):
        _need(w_, xs[0].dtype, "weight"); _need(b_, torch.float32, "bias")
        assert tuple(w_.shape) == (cout, kh, kw, cin) and b_.numel() == cout
    for x in xs:
        _need(x, xs[0].dtype, "x")
        assert x.shape[3] == cin
    if outs is None:
        outs = [torch.empty(tuple(x.shape[:3]) + (cout,), dtype=x.dtype, device=x.device) for x in xs]
    for x, o in zip(xs, outs):
        _need(o, x.dtype, "out")
        assert tuple(o.shape) == tuple(x.shape[:3]) + (cout,)
    if FLOP_COUNT is not None:
        FLOP_COUNT["conv"] += sum(2.0 * x.shape[0] * x.shape[1] * x.shape[2] * cout * kh * kw * cin for x in xs)
    p = _levels_params(xs[0], w0, relu)
    arr = _conv_levels(xs, outs, weights=ws, biases=bs)
    st = lib.osr_conv2d_fwd_levels(C.byref(p), len(xs), arr, None if ws is not None else _p(weight), None if bs is not None else _p(bias), _stream())
    if st == _lib.ERR_UNSUPPORTED:
        return None
    check(st, "osr_conv2d_fwd_levels")
    return list(outs)


def cfrpn_head_fused_levels(xs: Sequence[torch.Tensor], weight: torch.Tensor, bias: torch.Tensor, w_tail: torch.Tensor, b_tail: torch.Tensor,
                            deltas_outs: Sequence[torch.Tensor], ctr_outs: Sequence[torch.Tensor],
                            hidden_outs: Optional[Sequence[Optional[torch.Tensor]]] = None) -> bool:
    """ClsFreeRPNHead.forward over all the given levels in ONE launch (osr_cfrpn_head_fwd_levels). deltas_outs[i] (n*h*w, 4) / ctr_outs[i]
    (n*h*w) fp32 receive level i's outputs (views of the level-major buffers rpn_select reads). Returns False -- nothing launched -- outside
    the fused kernel's envelope."""
    lib = _lib.load()
    cout, kh, kw, cin = weight.shape
    if cout != 256 or cin % 64 or kh != kw or kh % 2 == 0 or kh * kw * cin // 64 < 8 or xs[0].dtype not in (torch.float16, torch.bfloat16) or len(xs) > _lib.MAX_CONV_LEVELS:
        return False
    _need(weight, xs[0].dtype, "weight"); _need(bias, torch.float32, "bias")
    _need(w_tail, torch.float32, "w_tail"); _need(b_tail, torch.float32, "b_tail")
    for i, x in enumerate(xs):
        _need(x, xs[0].dtype, "x")
        rows = x.shape[0] * x.shape[1] * x.shape[2]
        _need(deltas_outs[i], torch.float32, "deltas"); _need(ctr_outs[i], torch.float32, "ctr")
        assert deltas_outs[i].numel() == rows * 4 and ctr_outs[i].numel() == rows and x.shape[3] == cin
        if hidden_outs is not None and hidden_outs[i] is not None:
            _need(hidden_outs[i], x.dtype, "hidden_out")
            assert hidden_outs[i].numel() == rows * 256
    if FLOP_COUNT is not None:
        FLOP_COUNT["conv"] += sum(2.0 * x.shape[0] * x.shape[1] * x.shape[2] * cout * (kh * kw * cin + 5) for x in xs)
    p = _levels_params(xs[0], weight, True)
    arr = _conv_levels(xs, hidden_outs, deltas_outs, ctr_outs)
    st = lib.osr_cfrpn_head_fwd_levels(C.byref(p), len(xs), arr, _p(weight), _p(bias), _p(w_tail), _p(b_tail), _stream())
    if st == _lib.ERR_UNSUPPORTED:
        return False
    check(st, "osr_cfrpn_head_fwd_levels")
    return True


def make_rpn_levels(shapes: Sequence[Tuple[int, int]], strides: Sequence[int], n: int, num_anchors: int = 1) -> RpnLevels:
    lv = RpnLevels()
    lv.num_levels, lv.num_anchors = len(shapes), num_anchors
    off = 0
    for i, ((h, w), s) in enumerate(zip(shapes, strides)):
        lv.h[i], lv.w[i], lv.stride[i], lv.offset[i] = h, w, s, off
        off += n * h * w * num_anchors
    return lv


def rpn_select(lv: RpnLevels, cell_anchors: torch.Tensor, ctr: torch.Tensor, deltas: torch.Tensor, n: int,
               image_hw: torch.Tensor, pre_nms_topk: int, min_box_size: float = 0.0, b2b_weights: Optional[Sequence[float]] = None):
    """ctr/deltas: level-major concatenation (see osr.h). Returns dict of padded outputs. b2b_weights: decode with
    [d2] Box2BoxTransform(weights) instead of the CF-RPN's ltrb rule and also return the pyramid level of every slot (the stock
    RPN of Base-RCNN-FPN.yaml, osr_rpn_select_ex)."""
    if b2b_weights is not None:
        return _rpn_select_ex(lv, cell_anchors, ctr, deltas, n, image_hw, pre_nms_topk, min_box_size, b2b_weights)
    lib = _lib.load()
    _need(cell_anchors, torch.float32, "cell_anchors"); _need(ctr, torch.float32, "ctr"); _need(deltas, torch.float32, "deltas")
    _need(image_hw, torch.int32, "image_hw")
    cap = lib.osr_rpn_select_capacity(C.byref(lv), pre_nms_topk)
    if cap < 0:
        check(cap, "osr_rpn_select_capacity")
    wsb = lib.osr_rpn_select_workspace_bytes(C.byref(lv), n, pre_nms_topk)
    dev = ctr.device
    ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
    boxes = torch.empty((n, cap, 4), dtype=torch.float32, device=dev)
    scores = torch.empty((n, cap), dtype=torch.float32, device=dev)
    src = torch.empty((n, cap), dtype=torch.int32, device=dev)
    bidx = torch.empty((n * cap,), dtype=torch.int32, device=dev)
    counts = torch.empty((n,), dtype=torch.int32, device=dev)
    flags = torch.zeros((1,), dtype=torch.int32, device=dev)
    check(lib.osr_rpn_select(C.byref(lv), _p(cell_anchors), _p(ctr), _p(deltas), n, _p(image_hw), pre_nms_topk, float(min_box_size),
                             _p(boxes), _p(scores), _p(src), _p(bidx), _p(counts), _p(flags), _p(ws), wsb, _stream()), "osr_rpn_select")
    return dict(boxes=boxes, scores=scores, src_index=src, batch_idx=bidx, counts=counts, status_flags=flags, cap=cap)


def _rpn_select_ex(lv, cell_anchors, ctr, deltas, n, image_hw, pre_nms_topk, min_box_size, weights):
    lib = _lib.load()
    _need(cell_anchors, torch.float32, "cell_anchors"); _need(ctr, torch.float32, "scores"); _need(deltas, torch.float32, "deltas")
    _need(image_hw, torch.int32, "image_hw")
    cap = lib.osr_rpn_select_capacity(C.byref(lv), pre_nms_topk)
    if cap < 0:
        check(cap, "osr_rpn_select_capacity")
    wsb = lib.osr_rpn_select_workspace_bytes(C.byref(lv), n, pre_nms_topk)
    dev = ctr.device
    ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
    boxes = torch.empty((n, cap, 4), dtype=torch.float32, device=dev)
    scores = torch.empty((n, cap), dtype=torch.float32, device=dev)
    src = torch.empty((n, cap), dtype=torch.int32, device=dev)
    level = torch.empty((n, cap), dtype=torch.int32, device=dev)
    bidx = torch.empty((n * cap,), dtype=torch.int32, device=dev)
    counts = torch.empty((n,), dtype=torch.int32, device=dev)
    flags = torch.zeros((1,), dtype=torch.int32, device=dev)
    rw = (C.c_float * 4)(*[float(v) for v in weights])
    check(lib.osr_rpn_select_ex(C.byref(lv), _p(cell_anchors), _p(ctr), _p(deltas), n, _p(image_hw), pre_nms_topk, float(min_box_size), 1, rw,
                                _p(boxes), _p(scores), _p(src), _p(bidx), _p(level), _p(counts), _p(flags), _p(ws), wsb, _stream()), "osr_rpn_select_ex")
    return dict(boxes=boxes, scores=scores, src_index=src, batch_idx=bidx, level=level, counts=counts, status_flags=flags, cap=cap)


def fastrcnn_candidates(logits, deltas, prop_boxes, prop_count, image_hw, num_classes: int, reg_weights=(10.0, 10.0, 5.0, 5.0),
                        score_thresh: float = 0.05):
    """[d2] fast_rcnn_inference_single_image before its NMS. logits (n*rows, K+1), deltas (n*rows, K*4 or 4), prop_boxes (n,rows,4)."""
    lib = _lib.load()
    _need(logits, torch.float32, "logits"); _need(deltas, torch.float32, "deltas"); _need(prop_boxes, torch.float32, "prop_boxes")
    _need(prop_count, torch.int32, "prop_count"); _need(image_hw, torch.int32, "image_hw")
    n, rows = prop_boxes.shape[0], prop_boxes.shape[1]
    kbox = deltas.shape[1] // 4
    if logits.shape != (n * rows, num_classes + 1) or deltas.shape[0] != n * rows or kbox not in (1, num_classes):
        raise OsrError(f"fastrcnn_candidates: logits {tuple(logits.shape)} / deltas {tuple(deltas.shape)} do not fit {n} x {rows} rows, {num_classes} classes")
    dev, cap = logits.device, rows * num_classes
    o = dict(boxes=torch.empty((n, cap, 4), dtype=torch.float32, device=dev), scores=torch.empty((n, cap), dtype=torch.float32, device=dev),
             cls=torch.empty((n, cap), dtype=torch.int32, device=dev), row=torch.empty((n, cap), dtype=torch.int32, device=dev),
             count=torch.empty((n,), dtype=torch.int32, device=dev), cap=cap)
    rw = (C.c_float * 4)(*[float(v) for v in reg_weights])
    check(lib.osr_fastrcnn_candidates(_p(logits), _p(deltas), num_classes, kbox, _p(prop_boxes), _p(prop_count), n, rows, _p(image_hw), rw,
                                      float(score_thresh), _p(o["boxes"]), _p(o["scores"]), _p(o["cls"]), _p(o["row"]), _p(o["count"]), _stream()),
          "osr_fastrcnn_candidates")
    return o


def _pyramid(feats, scales) -> "Pyramid":
    py = Pyramid()
    py.num_levels, py.c = len(feats), feats[0].shape[3]
    for i, (f, s) in enumerate(zip(feats, scales)):
        _need(f, feats[0].dtype, f"feats[{i}]")
        py.h[i], py.w[i], py.scale[i], py.data[i] = f.shape[1], f.shape[2], float(s), f.data_ptr()
    return py


# RoIAlign pools the RoIs in locality order (osr_roi_locality_order) unless told otherwise: same output bits, ~8x fewer re-reads
# from HBM than the score order the proposal lists arrive in.
ROI_LOCALITY_ORDER = True


def roi_locality_order(feats: List[torch.Tensor], scales: Sequence[float], boxes: torch.Tensor, batch_idx: torch.Tensor,
                       canonical_level: int = 4, canonical_size: int = 224, min_level: int = 2) -> torch.Tensor:
    """(m + 1,) int32: a permutation of the RoI list that puts RoIs of one image / level / 32-pixel tile next to each other, padding
    rows (batch index -1) last, followed by one more entry: the number of non-padding rows."""
    lib = _lib.load()
    _need(boxes, torch.float32, "boxes"); _need(batch_idx, torch.int32, "batch_idx")
    py = _pyramid(feats, scales)
    m, n = boxes.shape[0], feats[0].shape[0]
    order = torch.empty((m + 1,), dtype=torch.int32, device=boxes.device)
    nb = int(lib.osr_roi_locality_order_workspace_bytes(n, m))
    ws = torch.empty((max(nb, 4),), dtype=torch.uint8, device=boxes.device)
    check(lib.osr_roi_locality_order(C.byref(py), n, _p(boxes), _p(batch_idx), m, canonical_level, canonical_size, min_level, _p(order),
                                     C.c_void_p(order.data_ptr() + 4 * m), _p(ws), nb, _stream()), "osr_roi_locality_order")
    return order


def roi_align(feats: List[torch.Tensor], scales: Sequence[float], boxes: torch.Tensor, batch_idx: torch.Tensor,
              pooled: int = 7, out_dtype: Optional[torch.dtype] = None, canonical_level: int = 4, canonical_size: int = 224,
              min_level: int = 2, order: Optional[torch.Tensor] = None, fill_padding: bool = True) -> torch.Tensor:
    """feats: NHWC per level; boxes (m,4) fp32; batch_idx (m) int32. Returns (m, pooled, pooled, c). order: processing order, (m,)
    int32 or (m + 1,) as roi_locality_order returns it (None: that order when ROI_LOCALITY_ORDER, else list order); the result
    does not depend on it. fill_padding=False: the rows of padding entries (batch index -1) are left unwritten instead of zeroed."""
    lib = _lib.load()
    _need(boxes, torch.float32, "boxes"); _need(batch_idx, torch.int32, "batch_idx")
    py = _pyramid(feats, scales)
    m = boxes.shape[0]
    if order is None and ROI_LOCALITY_ORDER and m > 0:
        order = roi_locality_order(feats, scales, boxes, batch_idx, canonical_level, canonical_size, min_level)
    nvalid = None
    if order is not None:
        _need(order, torch.int32, "order")
        assert order.numel() in (m, m + 1)
        if order.numel() == m + 1:
            nvalid = C.c_void_p(order.data_ptr() + 4 * m)
    out_dtype = out_dtype or feats[0].dtype
    out = torch.empty((m, pooled, pooled, py.c), dtype=out_dtype, device=boxes.device)
    check(lib.osr_roi_align_fwd_ordered_ex(C.byref(py), _DT[feats[0].dtype], feats[0].shape[0], _p(boxes), _p(batch_idx), m, pooled,
                                           canonical_level, canonical_size, min_level, _p(order), nvalid, 0 if fill_padding else 1, _p(out),
                                           _DT[out_dtype], _stream()), "osr_roi_align_fwd")
    return out


def box_predictor_tail(x, w, b, proposals, ctr, batch_idx, image_hw, reg_weights=(10.0, 10.0, 5.0, 5.0), mean_type=0,
                       score_thresh=0.05):
    lib = _lib.load()
    for t, nme in ((x, "x"), (w, "w"), (b, "b"), (proposals, "proposals"), (ctr, "ctr")):
        _need(t, torch.float32, nme)
    _need(batch_idx, torch.int32, "batch_idx"); _need(image_hw, torch.int32, "image_hw")
    m, k = x.shape
    dev = x.device
    pd = torch.empty((m, 4), dtype=torch.float32, device=dev)
    pi = torch.empty((m,), dtype=torch.float32, device=dev)
    bx = torch.empty((m, 4), dtype=torch.float32, device=dev)
    sc = torch.empty((m,), dtype=torch.float32, device=dev)
    cd = torch.empty((m,), dtype=torch.int32, device=dev)
    rw = (C.c_float * 4)(*[float(v) for v in reg_weights])
    check(lib.osr_box_predictor_tail(_p(x), m, k, _p(w), _p(b), _p(proposals), _p(ctr), _p(batch_idx), _p(image_hw), rw, mean_type,
                                     float(score_thresh), _p(pd), _p(pi), _p(bx), _p(sc), _p(cd), _stream()), "osr_box_predictor_tail")
    return dict(pred_deltas=pd, pred_iou=pi, boxes=bx, score=sc, cand=cd)


def nms_topk(boxes, scores, cls, cand, num_segments: int, seg_stride: int, seg_len, thr: float, topk: int):
    lib = _lib.load()
    _need(boxes, torch.float32, "boxes"); _need(scores, torch.float32, "scores"); _need(seg_len, torch.int32, "seg_len")
    if cls is not None:
        _need(cls, torch.int32, "cls")
    if cand is not None:
        _need(cand, torch.int32, "cand")
    dev = boxes.device
    wsb = lib.osr_nms_topk_workspace_bytes(num_segments, seg_stride)
    if wsb < 0:
        check(int(wsb), "osr_nms_topk_workspace_bytes")
    ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
    keep = torch.empty((num_segments, topk), dtype=torch.int32, device=dev)
    cnt = torch.empty((num_segments,), dtype=torch.int32, device=dev)
    check(lib.osr_nms_topk(_p(boxes), _p(scores), _p(cls), _p(cand), num_segments, seg_stride, _p(seg_len), float(thr), topk, _p(keep),
                           _p(cnt), _p(ws), wsb, _stream()), "osr_nms_topk")
    return keep, cnt


def gather_rows(src, seg_stride: int, keep, keep_count) -> torch.Tensor:
    lib = _lib.load()
    _need(src, torch.float32, "src"); _need(keep, torch.int32, "keep"); _need(keep_count, torch.int32, "keep_count")
    nseg, topk = keep.shape
    row = src.shape[-1] if src.dim() > 1 else 1
    dst = torch.empty((nseg, topk, row), dtype=torch.float32, device=src.device)
    check(lib.osr_gather_rows(_p(src), seg_stride, row, _p(keep), _p(keep_count), nseg, topk, _p(dst), _stream()), "osr_gather_rows")
    return dst


def l2_normalize_rows(x: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    _need(x, torch.float32, "x")
    out = torch.empty_like(x)
    check(lib.osr_l2_normalize_rows(_p(x), x.shape[0], x.shape[1], _p(out), _stream()), "osr_l2_normalize_rows")
    return out


def _pln_distance(name: str) -> int:
    if name not in _lib.PLN_DISTANCES:
        raise OsrError(f"MODEL.PLN.DISTANCE_TYPE '{name}': one of {sorted(_lib.PLN_DISTANCES)}")
    return _lib.PLN_DISTANCES[name]


def pln_tail(emb, protos_normed, num_known: int, reps: int, unk_thr: float, unknown_id: int, class_map=None, rows_valid=None,
             seg_rows: int = 0, distance: str = "COS"):
    lib = _lib.load()
    _need(emb, torch.float32, "emb"); _need(protos_normed, torch.float32, "protos")
    if class_map is not None:
        _need(class_map, torch.int64, "class_map")
    if rows_valid is not None:
        _need(rows_valid, torch.int32, "rows_valid")
    rows, d = emb.shape
    pc = torch.empty((rows,), dtype=torch.int64, device=emb.device)
    md = torch.empty((rows,), dtype=torch.float32, device=emb.device)
    check(lib.osr_pln_tail_ex(_p(emb), rows, d, _p(protos_normed), num_known, reps, _pln_distance(distance), float(unk_thr), int(unknown_id),
                              _p(class_map), _p(rows_valid), seg_rows, _p(pc), _p(md), _stream()), "osr_pln_tail")
    return pc, md


def softmax_candidates(logits, num_known, det_boxes, det_scores, pred_class, det_count, n, seg_rows, unknown_id, known_thresh,
                       unknown_thresh):
    lib = _lib.load()
    _need(logits, torch.float32, "logits"); _need(det_boxes, torch.float32, "det_boxes"); _need(det_scores, torch.float32, "det_scores")
    _need(pred_class, torch.int64, "pred_class"); _need(det_count, torch.int32, "det_count")
    dev = logits.device
    kcap = seg_rows * num_known
    o = dict(
        k_boxes=torch.empty((n, kcap, 4), dtype=torch.float32, device=dev), k_scores=torch.empty((n, kcap), dtype=torch.float32, device=dev),
        k_cls=torch.empty((n, kcap), dtype=torch.int32, device=dev), k_det=torch.empty((n, kcap), dtype=torch.int32, device=dev),
        k_count=torch.empty((n,), dtype=torch.int32, device=dev),
        u_boxes=torch.empty((n, seg_rows, 4), dtype=torch.float32, device=dev), u_scores=torch.empty((n, seg_rows), dtype=torch.float32, device=dev),
        u_det=torch.empty((n, seg_rows), dtype=torch.int32, device=dev), u_count=torch.empty((n,), dtype=torch.int32, device=dev))
    check(lib.osr_softmax_candidates(_p(logits), num_known, _p(det_boxes), _p(det_scores), _p(pred_class), _p(det_count), n, seg_rows,
                                     int(unknown_id), float(known_thresh), float(unknown_thresh), _p(o["k_boxes"]), _p(o["k_scores"]),
                                     _p(o["k_cls"]), _p(o["k_det"]), _p(o["k_count"]), _p(o["u_boxes"]), _p(o["u_scores"]), _p(o["u_det"]),
                                     _p(o["u_count"]), _stream()), "osr_softmax_candidates")
    return o


def assemble_detections(cands, k_keep, k_keep_count, u_keep, u_keep_count, n, unknown_id, class_map=None):
    lib = _lib.load()
    k_topk, u_topk = k_keep.shape[1], u_keep.shape[1]
    dev = k_keep.device
    cap = k_topk + u_topk
    ob = torch.empty((n, cap, 4), dtype=torch.float32, device=dev)
    os_ = torch.empty((n, cap), dtype=torch.float32, device=dev)
    oc = torch.empty((n, cap), dtype=torch.int64, device=dev)
    on = torch.empty((n,), dtype=torch.int32, device=dev)
    check(lib.osr_assemble_detections(_p(cands["k_boxes"]), _p(cands["k_scores"]), _p(cands["k_cls"]), _p(k_keep), _p(k_keep_count),
                                      cands["k_boxes"].shape[1], k_topk, _p(cands["u_boxes"]), _p(cands["u_scores"]), _p(u_keep),
                                      _p(u_keep_count), cands["u_boxes"].shape[1], u_topk, n, int(unknown_id), _p(class_map), _p(ob),
                                      _p(os_), _p(oc), _p(on), _stream()), "osr_assemble_detections")
    return ob, os_, oc, on


def detector_postprocess(boxes, scores, classes, count, scale_xy, out_hw):
    """Padded (n,cap,.) detections -> rescaled, clipped, empties dropped (order kept); scale_xy (n,2) fp32, out_hw (n,2) int32."""
    lib = _lib.load()
    _need(boxes, torch.float32, "boxes"); _need(scores, torch.float32, "scores"); _need(classes, torch.int64, "classes")
    _need(count, torch.int32, "count"); _need(scale_xy, torch.float32, "scale_xy"); _need(out_hw, torch.int32, "out_hw")
    n, cap = scores.shape
    ob, os_, oc = torch.empty_like(boxes), torch.empty_like(scores), torch.empty_like(classes)
    on = torch.empty_like(count)
    check(lib.osr_detector_postprocess(_p(boxes), _p(scores), _p(classes), _p(count), n, cap, _p(scale_xy), _p(out_hw), _p(ob), _p(os_), _p(oc),
                                       _p(on), _stream()), "osr_detector_postprocess")
    return ob, os_, oc, on


# ----------------------------------------------------------------------------------------------------------
# training step, forward half (targets + losses)
# ----------------------------------------------------------------------------------------------------------
def _levels_r(lv: RpnLevels) -> int:
    return sum(lv.h[i] * lv.w[i] for i in range(lv.num_levels)) * lv.num_anchors


def rpn_match_anchors(lv: RpnLevels, cell_anchors, n: int, gt_boxes, gt_count, reg_thr=(0.3, 0.7), obj_thr=(0.1, 0.3)):
    """gt_boxes (n,gmax,4) fp32 padded, gt_count (n) int32. Returns matched_idx, matched_iou, labels_reg, labels_obj (n,R)."""
    lib = _lib.load()
    _need(cell_anchors, torch.float32, "cell_anchors"); _need(gt_boxes, torch.float32, "gt_boxes"); _need(gt_count, torch.int32, "gt_count")
    gmax, r, dev = gt_boxes.shape[1], _levels_r(lv), gt_boxes.device
    midx = torch.empty((n, r), dtype=torch.int32, device=dev)
    miou = torch.empty((n, r), dtype=torch.float32, device=dev)
    lr = torch.empty((n, r), dtype=torch.int8, device=dev)
    lo = torch.empty((n, r), dtype=torch.int8, device=dev)
    ws = torch.empty((n * gmax * 4,), dtype=torch.uint8, device=dev)
    check(lib.osr_rpn_match_anchors(C.byref(lv), _p(cell_anchors), n, _p(gt_boxes), _p(gt_count), gmax, reg_thr[0], reg_thr[1], obj_thr[0],
                                    obj_thr[1], _p(midx), _p(miou), _p(lr), _p(lo), _p(ws), ws.numel(), _stream()), "osr_rpn_match_anchors")
    return midx, miou, lr, lo


def subsample_labels_(labels, keys, num_samples: int, positive_fraction: float):
    """In place on labels (n,r) int8; keys (n,r) fp32 uniform. Returns (num_pos, num_neg) int32 (n)."""
    lib = _lib.load()
    _need(labels, torch.int8, "labels"); _need(keys, torch.float32, "keys")
    if labels.shape != keys.shape or labels.dim() != 2:
        raise OsrError("labels and keys must both be (n, r)")
    n, r = labels.shape
    npos = torch.empty((n,), dtype=torch.int32, device=labels.device)
    nneg = torch.empty((n,), dtype=torch.int32, device=labels.device)
    check(lib.osr_subsample_labels(_p(labels), _p(keys), n, r, num_samples, positive_fraction, _p(npos), _p(nneg), _stream()),
          "osr_subsample_labels")
    return npos, nneg


def rpn_anchor_targets(lv: RpnLevels, cell_anchors, n: int, gt_boxes, gt_count, matched_idx, labels_obj):
    lib = _lib.load()
    _need(matched_idx, torch.int32, "matched_idx"); _need(labels_obj, torch.int8, "labels_obj")
    r, dev = _levels_r(lv), gt_boxes.device
    mb = torch.empty((n, r, 4), dtype=torch.float32, device=dev)
    ct = torch.empty((n, r), dtype=torch.float32, device=dev)
    check(lib.osr_rpn_anchor_targets(C.byref(lv), _p(cell_anchors), n, _p(gt_boxes), _p(gt_count), gt_boxes.shape[1], _p(matched_idx),
                                     _p(labels_obj), _p(mb), _p(ct), _stream()), "osr_rpn_anchor_targets")
    return mb, ct


def _loss_options(box, aux_beta):
    """box = (type name, smooth-L1 beta) as the yaml names it; aux_beta: beta of the centerness / IoU smooth-L1 loss."""
    o = _lib.LossOptions()
    if box[0] not in _lib.BOX_LOSS_TYPES:
        raise OsrError(f"box regression loss type '{box[0]}': one of {sorted(_lib.BOX_LOSS_TYPES)}")
    o.box_loss_type, o.box_smooth_l1_beta, o.aux_smooth_l1_beta = _lib.BOX_LOSS_TYPES[box[0]], float(box[1]), float(aux_beta)
    return o


def rpn_losses_fwd(lv: RpnLevels, cell_anchors, n: int, pred_deltas, pred_ctr, labels_reg, labels_obj, matched_boxes, ctr_target,
                   loc_weight=0.5, ctr_weight=0.5, batch_size_per_image=256, box_loss=("iou", 0.0), ctr_beta=0.0) -> torch.Tensor:
    """pred_*: level-major (as the RPN head writes them). Returns 6 floats: loss_rpn_loc, loss_rpn_ctr, 4 anchor counts.
    box_loss: (MODEL.RPN.BBOX_REG_LOSS_TYPE, MODEL.RPN.SMOOTH_L1_BETA); ctr_beta: MODEL.RPN.CTR_SMOOTH_L1_BETA."""
    lib = _lib.load()
    _need(pred_deltas, torch.float32, "pred_deltas"); _need(pred_ctr, torch.float32, "pred_ctr")
    _need(matched_boxes, torch.float32, "matched_boxes"); _need(ctr_target, torch.float32, "ctr_target")
    dev = pred_ctr.device
    out = torch.empty((6,), dtype=torch.float32, device=dev)
    ws = torch.empty((256 * 6 * 4,), dtype=torch.uint8, device=dev)
    opt = _loss_options(box_loss, ctr_beta)
    check(lib.osr_rpn_losses_fwd_ex(C.byref(lv), _p(cell_anchors), n, _p(pred_deltas), _p(pred_ctr), _p(labels_reg), _p(labels_obj),
                                    _p(matched_boxes), _p(ctr_target), loc_weight, ctr_weight, batch_size_per_image, C.byref(opt), _p(out), _p(ws),
                                    ws.numel(), _stream()), "osr_rpn_losses_fwd")
    return out


def roi_match_and_sample(prop_boxes, prop_logits, prop_count, gt_boxes, gt_classes, gt_count, keys, num_classes: int,
                         batch_size: int = 512, positive_fraction: float = 0.25, iou_thr: float = 0.5):
    """prop_boxes (n,pcap,4), prop_logits (n,pcap), prop_count (n) int32, gt_boxes (n,gmax,4), gt_classes (n,gmax) int64,
    gt_count (n) int32, keys (n,pcap+gmax). Returns a dict of padded (n,batch_size,...) outputs."""
    lib = _lib.load()
    _need(prop_boxes, torch.float32, "prop_boxes"); _need(prop_logits, torch.float32, "prop_logits"); _need(prop_count, torch.int32, "prop_count")
    _need(gt_boxes, torch.float32, "gt_boxes"); _need(gt_classes, torch.int64, "gt_classes"); _need(gt_count, torch.int32, "gt_count")
    _need(keys, torch.float32, "keys")
    n, pcap = prop_boxes.shape[0], prop_boxes.shape[1]
    gmax, dev = gt_boxes.shape[1], prop_boxes.device
    if tuple(keys.shape) != (n, pcap + gmax):
        raise OsrError(f"keys must be (n, pcap+gmax) = ({n}, {pcap + gmax}), got {tuple(keys.shape)}")
    wsb = lib.osr_roi_match_sample_workspace_bytes(n, pcap, gmax)
    if wsb < 0:
        check(int(wsb), "osr_roi_match_sample_workspace_bytes")
    ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
    o = dict(boxes=torch.empty((n, batch_size, 4), dtype=torch.float32, device=dev),
             logits=torch.empty((n, batch_size), dtype=torch.float32, device=dev),
             gt_classes=torch.empty((n, batch_size), dtype=torch.int64, device=dev),
             ious=torch.empty((n, batch_size), dtype=torch.float32, device=dev),
             gt_boxes=torch.empty((n, batch_size, 4), dtype=torch.float32, device=dev),
             src=torch.empty((n, batch_size), dtype=torch.int32, device=dev),
             batch_idx=torch.empty((n * batch_size,), dtype=torch.int32, device=dev),
             counts=torch.empty((n, 3), dtype=torch.int32, device=dev))
    check(lib.osr_roi_match_and_sample(_p(prop_boxes), _p(prop_logits), _p(prop_count), pcap, _p(gt_boxes), _p(gt_classes), _p(gt_count), gmax, n,
                                       _p(keys), num_classes, batch_size, positive_fraction, iou_thr, _p(o["boxes"]), _p(o["logits"]),
                                       _p(o["gt_classes"]), _p(o["ious"]), _p(o["gt_boxes"]), _p(o["src"]), _p(o["batch_idx"]), _p(o["counts"]), _p(ws), wsb,
                                       _stream()), "osr_roi_match_and_sample")
    return o


def _loss_ws(dev, nv: int):
    return torch.empty((256 * nv * 4,), dtype=torch.uint8, device=dev)


def roi_box_losses_fwd(pred_deltas, pred_iou, proposal_boxes, gt_boxes, gt_classes, gt_iou, num_classes: int,
                       reg_weights=(10.0, 10.0, 5.0, 5.0), box_weight=0.5, iou_weight=0.5, iou_is_logit: bool = False,
                       box_loss=("smooth_l1", 0.0), iou_beta=0.0) -> torch.Tensor:
    """pred_deltas (m,>=4) / pred_iou (m) may be column views of one row-major predictor output (stride(1) == 1).
    Returns 3 floats: loss_box_reg, loss_iou, rows counted (class >= 0). box_loss: (MODEL.ROI_BOX_HEAD.BBOX_REG_LOSS_TYPE,
    SMOOTH_L1_BETA); iou_beta: IOU_SMOOTH_L1_BETA."""
    lib = _lib.load()
    for t, nm in ((proposal_boxes, "proposal_boxes"), (gt_boxes, "gt_boxes"), (gt_iou, "gt_iou")):
        _need(t, torch.float32, nm)
    _need(gt_classes, torch.int64, "gt_classes")
    for t, nm in ((pred_deltas, "pred_deltas"), (pred_iou, "pred_iou")):
        if not t.is_cuda or t.dtype != torch.float32 or (t.dim() == 2 and t.stride(1) != 1):
            raise OsrError(f"{nm} must be an fp32 GPU tensor with unit stride inside a row")
    m, dev = gt_classes.numel(), proposal_boxes.device
    out = torch.empty((3,), dtype=torch.float32, device=dev)
    ws = _loss_ws(dev, 3)
    rw = (C.c_float * 4)(*reg_weights)
    opt = _loss_options(box_loss, iou_beta)
    check(lib.osr_roi_box_losses_fwd_ex(_p(pred_deltas), pred_deltas.stride(0), _p(pred_iou), pred_iou.stride(0), int(iou_is_logit), _p(proposal_boxes),
                                        _p(gt_boxes), _p(gt_classes), _p(gt_iou), m, num_classes, rw, box_weight, iou_weight, C.byref(opt), _p(out), _p(ws),
                                        ws.numel(), _stream()), "osr_roi_box_losses_fwd")
    return out


def pln_loss_fwd(emb, protos_normed, gt_classes, ious, iou_thr: float, alpha: float, beta: float, loss_weight: float, reps: int = 1,
                 distance: str = "COS") -> torch.Tensor:
    """protos_normed: (num_known * reps, d), the reps prototypes of a class next to each other (prototype_learning_network.py:163)."""
    lib = _lib.load()
    _need(emb, torch.float32, "emb"); _need(protos_normed, torch.float32, "protos_normed"); _need(gt_classes, torch.int64, "gt_classes")
    _need(ious, torch.float32, "ious")
    m, d = emb.shape
    out = torch.empty((1,), dtype=torch.float32, device=emb.device)
    ws = _loss_ws(emb.device, 4)
    assert protos_normed.shape[0] % reps == 0
    check(lib.osr_pln_loss_fwd_ex(_p(emb), m, d, _p(protos_normed), protos_normed.shape[0] // reps, reps, _pln_distance(distance), _p(gt_classes),
                                  _p(ious), iou_thr, alpha, beta, loss_weight, _p(out), _p(ws), ws.numel(), _stream()), "osr_pln_loss_fwd")
    return out


def softmax_ce_loss_fwd(logits, gt_classes, num_classes: int, loss_weight: float) -> torch.Tensor:
    lib = _lib.load()
    _need(logits, torch.float32, "logits"); _need(gt_classes, torch.int64, "gt_classes")
    m, nk1 = logits.shape
    out = torch.empty((1,), dtype=torch.float32, device=logits.device)
    ws = _loss_ws(logits.device, 2)
    check(lib.osr_softmax_ce_loss_fwd(_p(logits), m, nk1 - 1, _p(gt_classes), num_classes, loss_weight, _p(out), _p(ws), ws.numel(), _stream()),
          "osr_softmax_ce_loss_fwd")
    return out


# ----------------------------------------------------------------------------------------------------------
# training step, backward half: dense layers
# ----------------------------------------------------------------------------------------------------------
def _conv_params(n, hi, wi, cin, ho, wo, cout, kh, kw, stride, pad, dt_in, dt_out) -> ConvParams:
    p = _new_conv_params()
    p.n, p.hi, p.wi, p.cin, p.ho, p.wo, p.cout = n, hi, wi, cin, ho, wo, cout
    p.kh, p.kw, p.stride_h, p.stride_w, p.pad_h, p.pad_w = kh, kw, stride, stride, pad, pad
    p.in_stride_n, p.in_stride_h, p.in_stride_w = hi * wi * cin, wi * cin, cin
    p.out_stride_n, p.out_stride_h, p.out_stride_w = ho * wo * cout, wo * cout, cout
    p.relu, p.res_mode, p.pad_mode = 0, 0, 0
    p.in_dtype, p.out_dtype = _DT[dt_in], _DT[dt_out]
    return p


def conv2d_dgrad(dy: torch.Tensor, w_dgrad: torch.Tensor, x_hw: Tuple[int, int], stride: int = 1, pad: int = 0,
                 mask: Optional[torch.Tensor] = None, add: Optional[torch.Tensor] = None, out_dtype: Optional[torch.dtype] = None,
                 post_mask: Optional[torch.Tensor] = None, strided_only: bool = False) -> torch.Tensor:
    """Gradient w.r.t. the input of a convolution: dy (n,ho,wo,cout), w_dgrad = pack_dgrad_weight(w) (cin,kh,kw,cout).
    mask: forward activation at dx's positions (n,hi,wi,cin) -> dx is zeroed where mask <= 0 (the ReLU below);
    add: a second gradient of dx's shape summed in (residual / shortcut branch). At most one of the two (they share the
    epilogue's auxiliary operand); post_mask: a forward activation of dx's shape applied as a ReLU mask AFTER the sum (the
    join of a residual block: osr_conv2d_fwd_masked); stride 2 is supported for 1x1 layers (every second pixel of a zeroed dx).
    strided_only (stride > 1): the caller reads dx only at the pixels the stride visits (as the `add` of a second strided launch
    does), so the other pixels are left unwritten instead of zero-filled."""
    lib = _lib.load()
    _need(dy, name="dy"); _need(w_dgrad, dy.dtype, "w_dgrad")
    n, ho, wo, cout = dy.shape
    cin, kh, kw, cout2 = w_dgrad.shape
    hi, wi = x_hw
    if cout2 != cout:
        raise OsrError(f"w_dgrad cout {cout2} != dy channels {cout}")
    if mask is not None and add is not None:
        raise OsrError("conv2d_dgrad takes a mask or an addend, not both")
    aux, mode = (mask, 3) if mask is not None else ((add, 1) if add is not None else (None, 0))
    out_dtype = out_dtype or dy.dtype
    zero_bias = _zero_bias(cin, dy.device)
    if stride == 1:
        if (hi + 2 * pad - kh) + 1 != ho or (wi + 2 * pad - kw) + 1 != wo:
            raise OsrError("x_hw inconsistent with dy and the kernel geometry")
        return conv2d(dy, w_dgrad, zero_bias, 1, kh - 1 - pad, False, aux, mode, out_dtype, post_mask=post_mask)
    if not (kh == 1 and kw == 1 and pad == 0):
        raise OsrError("strided backward-data is implemented for 1x1 layers only (the reference's R-50 strides in the 1x1)")
    if (hi - 1) // stride + 1 != ho or (wi - 1) // stride + 1 != wo:
        raise OsrError("x_hw inconsistent with dy and the stride")
    if FLOP_COUNT is not None:
        FLOP_COUNT["conv"] += 2.0 * n * ho * wo * cout * cin
    dx = (torch.empty if strided_only else torch.zeros)((n, hi, wi, cin), dtype=out_dtype, device=dy.device)
    p = _conv_params(n, ho, wo, cout, ho, wo, cin, 1, 1, 1, 0, dy.dtype, out_dtype)
    p.out_stride_n, p.out_stride_h, p.out_stride_w = hi * wi * cin, stride * wi * cin, stride * cin
    p.res_mode = mode
    if aux is not None:
        _need(aux, dy.dtype, "mask/add")
        if tuple(aux.shape) != (n, hi, wi, cin):
            raise OsrError(f"mask/add shape {tuple(aux.shape)} != {(n, hi, wi, cin)}")
        p.res_stride_n, p.res_stride_h, p.res_stride_w = hi * wi * cin, stride * wi * cin, stride * cin
    if post_mask is not None:
        _need(post_mask, dy.dtype, "post_mask")
        if tuple(post_mask.shape) != (n, hi, wi, cin):
            raise OsrError(f"post_mask shape {tuple(post_mask.shape)} != {(n, hi, wi, cin)}")
        if mode == 3:
            raise OsrError("conv2d_dgrad: mask and post_mask are the same thing; pass one")
        st = lib.osr_conv2d_fwd_masked(C.byref(p), _p(dy), _p(w_dgrad), _p(zero_bias), _p(aux), _p(post_mask), _p(dx), _stream())
        if st != _lib.ERR_UNSUPPORTED:
            check(st, "osr_conv2d_fwd_masked(dgrad)")
            return dx  # (the pixels the stride skips stay zero: masking them changes nothing)
    check(lib.osr_conv2d_fwd(C.byref(p), _p(dy), _p(w_dgrad), _p(zero_bias), _p(aux), _p(dx), _stream()), "osr_conv2d_fwd(dgrad)")
    if post_mask is not None:
        relu_mask_(dx, post_mask)
    return dx


def conv2d_wgrad(x: torch.Tensor, dy: torch.Tensor, kh: int, kw: int, stride: int = 1, pad: int = 0, dw: Optional[torch.Tensor] = None,
                 accumulate: bool = False) -> torch.Tensor:
    """Weight gradient in the packed forward layout (cout,kh,kw,cin), fp32. x (n,hi,wi,cin), dy (n,ho,wo,cout) f16/bf16."""
    lib = _lib.load()
    _need(x, name="x"); _need(dy, x.dtype, "dy")
    n, hi, wi, cin = x.shape
    n2, ho, wo, cout = dy.shape
    if n2 != n or (hi + 2 * pad - kh) // stride + 1 != ho or (wi + 2 * pad - kw) // stride + 1 != wo:
        raise OsrError("dy shape inconsistent with x and the kernel geometry")
    if FLOP_COUNT is not None:
        FLOP_COUNT["wgrad"] += 2.0 * n * ho * wo * cout * kh * kw * cin
    p = _conv_params(n, hi, wi, cin, ho, wo, cout, kh, kw, stride, pad, x.dtype, x.dtype)
    if dw is None:
        dw = torch.empty((cout, kh, kw, cin), dtype=torch.float32, device=x.device)
        accumulate = False
    else:
        _need(dw, torch.float32, "dw")
    wsb = lib.osr_conv2d_wgrad_workspace_bytes(C.byref(p))
    if wsb < 0:
        check(int(wsb), "osr_conv2d_wgrad_workspace_bytes")
    ws = torch.empty((wsb,), dtype=torch.uint8, device=x.device)
    check(lib.osr_conv2d_wgrad(C.byref(p), _p(x), _p(dy), _p(dw), int(accumulate), _p(ws), wsb, _stream()), "osr_conv2d_wgrad")
    return dw


def linear_wgrad(x: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    """dW (n_out, k) fp32 of a fully connected layer: x (m,k), dy (m,n_out)."""
    m, k = x.shape
    return conv2d_wgrad(x.view(1, m, 1, k), dy.view(1, m, 1, dy.shape[1]), 1, 1).view(dy.shape[1], k)


def bias_grad(dy: torch.Tensor, db: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    """db[c] = sum over all leading dimensions of dy[..., c] (fp32)."""
    lib = _lib.load()
    _need(dy, name="dy")
    cout = dy.shape[-1]
    m = dy.numel() // cout
    if db is None:
        db = torch.empty((cout,), dtype=torch.float32, device=dy.device)
        accumulate = False
    ws = torch.empty((512 * cout * 4,), dtype=torch.uint8, device=dy.device)
    check(lib.osr_bias_grad(_p(dy), _DT[dy.dtype], m, cout, _p(db), int(accumulate), _p(ws), ws.numel(), _stream()), "osr_bias_grad")
    return db


# ----------------------------------------------------------------------------------------------------------
# training step, backward half: losses, per-row stages, elementwise, optimiser
# ----------------------------------------------------------------------------------------------------------
def rpn_losses_bwd(lv: RpnLevels, cell_anchors, n: int, pred_deltas, pred_ctr, labels_reg, labels_obj, matched_boxes, ctr_target,
                   loc_weight=0.5, ctr_weight=0.5, batch_size_per_image=256, loss_scale=1.0, box_loss=("iou", 0.0), ctr_beta=0.0) -> torch.Tensor:
    """-> d_out5 (rows, 5) fp32, level-major: gradient w.r.t. the head's {4 deltas, centerness logit}."""
    lib = _lib.load()
    out = torch.empty((pred_ctr.numel(), 5), dtype=torch.float32, device=pred_ctr.device)
    opt = _loss_options(box_loss, ctr_beta)
    check(lib.osr_rpn_losses_bwd_ex(C.byref(lv), _p(cell_anchors), n, _p(pred_deltas), _p(pred_ctr), _p(labels_reg), _p(labels_obj),
                                    _p(matched_boxes), _p(ctr_target), loc_weight, ctr_weight, batch_size_per_image, loss_scale, C.byref(opt), _p(out),
                                    _stream()), "osr_rpn_losses_bwd")
    return out


def cfrpn_tail_bwd(t: torch.Tensor, w_tail: torch.Tensor, d_out5: torch.Tensor):
    """t (rows,256) f16/bf16, w_tail (5,256) fp32, d_out5 (rows,5) fp32 -> dt (rows,256) like t, dw_tail (5,256), db_tail (5)."""
    lib = _lib.load()
    _need(t, name="t"); _need(w_tail, torch.float32, "w_tail"); _need(d_out5, torch.float32, "d_out5")
    rows = t.shape[0]
    dt = torch.empty_like(t)
    dw = torch.empty((5, 256), dtype=torch.float32, device=t.device)
    db = torch.empty((5,), dtype=torch.float32, device=t.device)
    wsb = lib.osr_cfrpn_tail_bwd_workspace_bytes()
    ws = torch.empty((wsb,), dtype=torch.uint8, device=t.device)
    check(lib.osr_cfrpn_tail_bwd(_p(t), _DT[t.dtype], rows, _p(w_tail), _p(d_out5), _p(dt), _p(dw), _p(db), 0, _p(ws), wsb, _stream()),
          "osr_cfrpn_tail_bwd")
    return dt, dw, db


def rpn_sparse_rows(d_out5: torch.Tensor, cap: int):
    """The rows of d_out5 (rows,5) with a non-zero gradient: (row_ids (cap) int32 ascending, -1 behind them; row_map (rows) int32 =
    list slot or -1; count2 = {listed, found})."""
    lib = _lib.load()
    _need(d_out5, torch.float32, "d_out5")
    rows = d_out5.shape[0]
    dev = d_out5.device
    ids = torch.empty((cap,), dtype=torch.int32, device=dev)
    rmap = torch.empty((rows,), dtype=torch.int32, device=dev)
    cnt = torch.empty((2,), dtype=torch.int32, device=dev)
    wsb = lib.osr_rpn_sparse_rows_workspace_bytes()
    ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
    check(lib.osr_rpn_sparse_rows(_p(d_out5), rows, cap, _p(ids), _p(rmap), _p(cnt), _p(ws), wsb, _stream()), "osr_rpn_sparse_rows")
    return ids, rmap, cnt


def rpn_gather_cols(lv: RpnLevels, feats: List[torch.Tensor], n: int, row_ids: torch.Tensor, d_out5: torch.Tensor):
    """-> (cols (cap, 2304) in the feature dtype: the im2col rows (tap-major) of the listed anchors; d_out5_rows (cap, 5))."""
    lib = _lib.load()
    _need(row_ids, torch.int32, "row_ids"); _need(d_out5, torch.float32, "d_out5")
    py = _pyramid(feats, [1.0] * len(feats))
    cap = row_ids.shape[0]
    cols = torch.empty((cap, 9 * 256), dtype=feats[0].dtype, device=row_ids.device)
    d5r = torch.empty((cap, 5), dtype=torch.float32, device=row_ids.device)
    check(lib.osr_rpn_gather_cols(C.byref(lv), C.byref(py), _DT[feats[0].dtype], n, _p(row_ids), cap, _p(d_out5), _p(cols), _p(d5r), _stream()),
          "osr_rpn_gather_cols")
    return cols, d5r


def rpn_scatter_cols_add_(lv: RpnLevels, n: int, row_map: torch.Tensor, y: torch.Tensor, grads: List[torch.Tensor]) -> List[torch.Tensor]:
    """In place on grads[l] (n, h_l, w_l, 256): += the per-tap data gradients y (cap, 2304) fp32 of the listed anchors (col2im)."""
    lib = _lib.load()
    _need(row_map, torch.int32, "row_map"); _need(y, torch.float32, "y")
    for i, gr in enumerate(grads):
        _need(gr, grads[0].dtype, f"grads[{i}]")
        if gr.shape[0] != n or gr.shape[1] != lv.h[i] or gr.shape[2] != lv.w[i] or gr.shape[3] != 256:
            raise OsrError(f"grads[{i}] must be ({n}, {lv.h[i]}, {lv.w[i]}, 256)")
    if len(grads) != lv.num_levels or y.shape[1] != 9 * 256:
        raise OsrError("one gradient tensor per level; y must be (cap, 2304)")
    ptrs = (C.c_void_p * len(grads))(*[gr.data_ptr() for gr in grads])
    check(lib.osr_rpn_scatter_cols_add(C.byref(lv), n, _p(row_map), _p(y), ptrs, _DT[grads[0].dtype], _stream()), "osr_rpn_scatter_cols_add")
    return grads


def roi_box_losses_bwd(pred5, proposal_boxes, gt_boxes, gt_classes, gt_iou, num_classes: int, reg_weights=(10.0, 10.0, 5.0, 5.0),
                       box_weight=0.5, iou_weight=0.5, loss_scale=1.0, box_loss=("smooth_l1", 0.0), iou_beta=0.0) -> torch.Tensor:
    lib = _lib.load()
    _need(pred5, torch.float32, "pred5")
    m = gt_classes.numel()
    out = torch.empty((m, 5), dtype=torch.float32, device=pred5.device)
    ws = torch.empty((16,), dtype=torch.uint8, device=pred5.device)
    rw = (C.c_float * 4)(*reg_weights)
    opt = _loss_options(box_loss, iou_beta)
    check(lib.osr_roi_box_losses_bwd_ex(_p(pred5), _p(proposal_boxes), _p(gt_boxes), _p(gt_classes), _p(gt_iou), m, num_classes, rw, box_weight,
                                        iou_weight, loss_scale, C.byref(opt), _p(out), _p(ws), 16, _stream()), "osr_roi_box_losses_bwd")
    return out


def softmax_ce_loss_bwd(logits, gt_classes, num_classes: int, loss_weight: float, loss_scale=1.0) -> torch.Tensor:
    lib = _lib.load()
    _need(logits, torch.float32, "logits")
    m, nk1 = logits.shape
    out = torch.empty_like(logits)
    ws = torch.empty((16,), dtype=torch.uint8, device=logits.device)
    check(lib.osr_softmax_ce_loss_bwd(_p(logits), m, nk1 - 1, _p(gt_classes), num_classes, loss_weight, loss_scale, _p(out), _p(ws), 16, _stream()),
          "osr_softmax_ce_loss_bwd")
    return out


def pln_loss_bwd(emb, protos_raw, gt_classes, ious, iou_thr: float, alpha: float, beta: float, loss_weight: float, loss_scale=1.0, reps: int = 1,
                 distance: str = "COS"):
    """-> (d_emb (m,d), d_protos (K * reps,d)) fp32; protos_raw are the un-normalised prototype parameters."""
    lib = _lib.load()
    _need(emb, torch.float32, "emb"); _need(protos_raw, torch.float32, "protos_raw")
    m, d = emb.shape
    de = torch.empty_like(emb)
    dp = torch.empty_like(protos_raw)
    wsb = lib.osr_pln_loss_bwd_workspace_bytes(m)
    ws = torch.empty((wsb,), dtype=torch.uint8, device=emb.device)
    assert protos_raw.shape[0] % reps == 0
    check(lib.osr_pln_loss_bwd_ex(_p(emb), m, d, _p(protos_raw), protos_raw.shape[0] // reps, reps, _pln_distance(distance), _p(gt_classes), _p(ious),
                                  iou_thr, alpha, beta, loss_weight, loss_scale, _p(de), _p(dp), 0, _p(ws), wsb, _stream()), "osr_pln_loss_bwd")
    return de, dp


def roi_align_bwd(dout: torch.Tensor, shapes: Sequence[Tuple[int, int]], n: int, scales: Sequence[float], boxes, batch_idx,
                  canonical_level: int = 4, canonical_size: int = 224, min_level: int = 2, rois_per_image: Optional[int] = None,
                  out_dtype: Optional[torch.dtype] = None) -> List[torch.Tensor]:
    """dout (m,P,P,c) -> list of (n,h,w,c) feature gradients, one per level: fp32, or out_dtype (= dout's dtype: the fp32 sums rounded
    once; the scatter path casts afterwards). rois_per_image: the list is image-major with this
    fixed stride (rows [b*S, (b+1)*S) are image b's or padding) -- the pixel-centric kernel then gathers (osr_roi_align_bwd_dense: no
    atomics, no zero fill, reproducible bit for bit); otherwise the scatter kernel adds into a zeroed pyramid with fp32 atomics."""
    lib = _lib.load()
    _need(dout, name="dout"); _need(boxes, torch.float32, "boxes"); _need(batch_idx, torch.int32, "batch_idx")
    m, pooled, _, c = dout.shape
    dense = rois_per_image is not None and m == n * rois_per_image and rois_per_image <= 1024 and c <= 256
    if out_dtype not in (None, torch.float32, dout.dtype):
        raise OsrError("roi_align_bwd: out_dtype must be float32 or dout's dtype")
    odt = out_dtype or torch.float32
    outs = [torch.empty((n, h, w, c), dtype=odt, device=dout.device) if dense else torch.zeros((n, h, w, c), dtype=torch.float32, device=dout.device)
            for h, w in shapes]
    py = Pyramid()
    py.num_levels, py.c = len(outs), c
    for i, (f, s) in enumerate(zip(outs, scales)):
        py.h[i], py.w[i], py.scale[i], py.data[i] = f.shape[1], f.shape[2], float(s), f.data_ptr()
    if dense:
        st = lib.osr_roi_align_bwd_dense(C.byref(py), n, _p(boxes), _p(batch_idx), m, int(rois_per_image), pooled, canonical_level, canonical_size,
                                         min_level, _p(dout), _DT[dout.dtype], _DT[odt], _stream())
        if st != _lib.ERR_UNSUPPORTED:
            check(st, "osr_roi_align_bwd_dense")
            return outs
        outs = [torch.zeros((n, h, w, c), dtype=torch.float32, device=dout.device) for h, w in shapes]
        for i, f in enumerate(outs):
            py.data[i] = f.data_ptr()
    check(lib.osr_roi_align_bwd(C.byref(py), n, _p(boxes), _p(batch_idx), m, pooled, canonical_level, canonical_size, min_level, _p(dout),
                                _DT[dout.dtype], _stream()), "osr_roi_align_bwd")
    return outs if odt == torch.float32 else [add_cast(f, None, odt) for f in outs]


def relu_mask_(g: torch.Tensor, act: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    _need(g, name="g"); _need(act, name="act")
    if g.numel() != act.numel():
        raise OsrError("relu_mask_: size mismatch")
    check(lib.osr_relu_mask(_p(g), _DT[g.dtype], _p(act), _DT[act.dtype], g.numel(), _stream()), "osr_relu_mask")
    return g


def add_cast(a_f32: Optional[torch.Tensor], b: Optional[torch.Tensor], dtype: torch.dtype) -> torch.Tensor:
    lib = _lib.load()
    ref = a_f32 if a_f32 is not None else b
    if a_f32 is not None:
        _need(a_f32, torch.float32, "a_f32")
    if b is not None:
        _need(b, dtype, "b")
    out = torch.empty(ref.shape, dtype=dtype, device=ref.device)
    check(lib.osr_add_cast(_p(a_f32), _p(b), _p(out), _DT[dtype], out.numel(), _stream()), "osr_add_cast")
    return out


def pool_bwd(src: torch.Tensor, out_hw: Tuple[int, int], base: Optional[torch.Tensor], mode: int) -> torch.Tensor:
    """mode 0: FPN top-down backward (out = base + 2x2 sums of src); mode 1: p6 subsample backward."""
    lib = _lib.load()
    _need(src, name="src")
    n, hs, ws_, c = src.shape
    ho, wo = out_hw
    if base is not None:
        _need(base, src.dtype, "base")
    out = torch.empty((n, ho, wo, c), dtype=src.dtype, device=src.device)
    check(lib.osr_pool_bwd(_p(src), hs, ws_, _p(base), _p(out), n, ho, wo, c, mode, _DT[src.dtype], _stream()), "osr_pool_bwd")
    return out


def pack_dgrad_weight(w: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(cout,kh,kw,cin) -> (cin,kh,kw,cout) spatially flipped (a 2-d (n,k) matrix is transposed to (k,n)); out: reuse this buffer."""
    lib = _lib.load()
    _need(w, name="w")
    if w.dim() == 2:
        cout, cin, kh, kw = w.shape[0], w.shape[1], 1, 1
        shape = (cin, cout)
    else:
        cout, kh, kw, cin = w.shape
        shape = (cin, kh, kw, cout)
    if out is None:
        out = torch.empty(shape, dtype=w.dtype, device=w.device)
    else:
        _need(out, w.dtype, "out")
        if out.numel() != w.numel():
            raise OsrError("pack_dgrad_weight: out has the wrong size")
    check(lib.osr_pack_dgrad_weight(_p(w), _p(out), cout, kh, kw, cin, _DT[w.dtype], _stream()), "osr_pack_dgrad_weight")
    return out


class MultiTensorPlan:
    """Device-resident table + chunk list of a multi-tensor launch (osr_sgd_step_multi / osr_pack_dgrad_weight_multi). `keep` holds the
    tensors the table points at: the plan is only valid while they live at the same addresses."""

    def __init__(self, table: torch.Tensor, chunks: torch.Tensor, num_chunks: int, keep, chunk_elems: int = 0):
        self.table, self.chunks, self.num_chunks, self.keep, self.chunk_elems = table, chunks, num_chunks, keep, chunk_elems


def _upload_struct_array(arr, device) -> torch.Tensor:
    raw = bytes(memoryview(arr).cast("B"))
    return torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)


SGD_CHUNK_ELEMS = 16384


def sgd_multi_plan(entries, device) -> MultiTensorPlan:
    """entries: [(param, grad, momentum_buf, row_scale | None, lowp | None)], fp32 param / grad / buf of equal size."""
    tab = (_lib.SgdTensor * len(entries))()
    chunks: List[int] = []
    keep = []
    for ti, (pm, gr, buf, rs, lp) in enumerate(entries):
        _need(pm, torch.float32, "param"); _need(gr, torch.float32, "grad"); _need(buf, torch.float32, "buf")
        if gr.numel() != pm.numel() or buf.numel() != pm.numel() or (lp is not None and lp.numel() != pm.numel()):
            raise OsrError("sgd_multi_plan: size mismatch")
        if rs is not None:
            _need(rs, torch.float32, "row_scale")
        if lp is not None:
            _need(lp, name="lowp")
        t = tab[ti]
        t.param, t.grad, t.momentum = pm.data_ptr(), gr.data_ptr(), buf.data_ptr()
        t.row_scale = rs.data_ptr() if rs is not None else None
        t.lowp = lp.data_ptr() if lp is not None else None
        t.n = pm.numel()
        t.row_elems = pm.numel() // pm.shape[0] if rs is not None else 1
        t.lowp_dtype = _DT[lp.dtype] if lp is not None else 0
        for c in range((pm.numel() + SGD_CHUNK_ELEMS - 1) // SGD_CHUNK_ELEMS):
            chunks += [ti, c]
        keep += [pm, gr, buf, rs, lp]
    ch = torch.tensor(chunks, dtype=torch.int32).to(device)
    return MultiTensorPlan(_upload_struct_array(tab, device), ch, len(chunks) // 2, keep, SGD_CHUNK_ELEMS)


def sgd_step_multi_(plan: MultiTensorPlan, lr: float, momentum: float, weight_decay: float, grad_scale: float = 1.0,
                    apply_flag: Optional[torch.Tensor] = None) -> None:
    """osr_sgd_step on every tensor of the plan, one launch."""
    lib = _lib.load()
    check(lib.osr_sgd_step_multi(_p(plan.table), _p(plan.chunks), plan.num_chunks, plan.chunk_elems, lr, momentum, weight_decay, grad_scale,
                                 _p(apply_flag), _stream()), "osr_sgd_step_multi")


def pack_dgrad_multi_plan(pairs, device) -> MultiTensorPlan:
    """pairs: [(w, out)] as pack_dgrad_weight takes them (w (cout,kh,kw,cin) or a 2-d (n,k) matrix; out of the same size and dtype)."""
    tab = (_lib.PackTensor * len(pairs))()
    chunks: List[int] = []
    keep = []
    for ti, (w, out) in enumerate(pairs):
        _need(w, name="w"); _need(out, w.dtype, "out")
        if out.numel() != w.numel() or w.element_size() not in (2, 4):
            raise OsrError("pack_dgrad_multi_plan: out has the wrong size / unsupported element size")
        cout, kh, kw, cin = (w.shape[0], 1, 1, w.shape[1]) if w.dim() == 2 else tuple(w.shape)
        t = tab[ti]
        t.src, t.dst, t.cout, t.kh, t.kw, t.cin, t.elem_bytes = w.data_ptr(), out.data_ptr(), cout, kh, kw, cin, w.element_size()
        for c in range(kh * kw * ((cout + 31) // 32) * ((cin + 31) // 32)):
            chunks += [ti, c]
        keep += [w, out]
    ch = torch.tensor(chunks, dtype=torch.int32).to(device)
    return MultiTensorPlan(_upload_struct_array(tab, device), ch, len(chunks) // 2, keep)


def pack_dgrad_weight_multi_(plan: MultiTensorPlan) -> None:
    """osr_pack_dgrad_weight on every pair of the plan, one launch."""
    lib = _lib.load()
    check(lib.osr_pack_dgrad_weight_multi(_p(plan.table), _p(plan.chunks), plan.num_chunks, _stream()), "osr_pack_dgrad_weight_multi")


_ZERO_BIAS = {}


def _zero_bias(n: int, device) -> torch.Tensor:
    """A shared all-zero fp32 bias (the data-gradient launches have none): allocated and filled once per (size, device)."""
    key = (n, str(device))
    if key not in _ZERO_BIAS:
        _ZERO_BIAS[key] = torch.zeros((n,), dtype=torch.float32, device=device)
    return _ZERO_BIAS[key]


def check_finite_(x: torch.Tensor, flag: torch.Tensor) -> torch.Tensor:
    """Clears flag (int32 (1,), preset to 1 by the caller) when x holds an inf / NaN. Asynchronous."""
    lib = _lib.load()
    _need(x, torch.float32, "x"); _need(flag, torch.int32, "flag")
    check(lib.osr_check_finite(_p(x), x.numel(), _p(flag), _stream()), "osr_check_finite")
    return flag


def sgd_step_(param, grad, buf, lr: float, momentum: float, weight_decay: float, grad_scale: float = 1.0, row_scale=None, lowp=None,
              apply_flag: Optional[torch.Tensor] = None):
    lib = _lib.load()
    _need(param, torch.float32, "param"); _need(grad, torch.float32, "grad"); _need(buf, torch.float32, "buf")
    if grad.numel() != param.numel() or buf.numel() != param.numel():
        raise OsrError("sgd_step_: size mismatch")
    row_elems = param.numel() // param.shape[0] if row_scale is not None else 1
    check(lib.osr_sgd_step(_p(param), _p(grad), _p(buf), param.numel(), lr, momentum, weight_decay, grad_scale, _p(row_scale), row_elems,
                           _p(lowp), _DT[lowp.dtype] if lowp is not None else 0, _p(apply_flag), _stream()), "osr_sgd_step")
