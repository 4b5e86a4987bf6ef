"""Thin torch-tensor wrappers over the C ABI (include/emadapt.h).

torch provides device memory and the current HIP stream; every computation happens inside
libemadapt_hip.so.  All tensors are fp32, NHWC, contiguous; filters are HWIO like the reference's
``tf.get_variable`` shapes (deeplab.py:131-149).
"""
import ctypes

import torch

from . import _lib
from ._lib import ConvDesc, EPI_BIAS, EPI_BIAS_RELU, EPI_NONE, LABEL_F32, LABEL_I32, LABEL_MULTIHOT  # noqa: F401


def _stream(t=None):
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _check(t, dtype=torch.float32, name="tensor"):
    if t is None:
        return
    if not t.is_cuda:
        raise ValueError("%s must live on the GPU (there is no CPU path)" % name)
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)


DT_F32, DT_BF16 = 0, 1  # EMADAPT_DT_*: storage type of an activation / activation-gradient tensor


def _act(t, name="tensor"):
    """Validate an activation tensor that may be stored f32 or bf16; returns its EMADAPT_DT_* code."""
    if t is None:
        return DT_F32
    if t.dtype == torch.float32:
        _check(t, torch.float32, name)
        return DT_F32
    if t.dtype == torch.bfloat16:
        _check(t, torch.bfloat16, name)
        if t.shape[-1] % 8 != 0:
            raise ValueError("%s: bf16 tensors need a channel count that is a multiple of 8" % name)
        return DT_BF16
    raise TypeError("%s must be float32 or bfloat16, got %s" % (name, t.dtype))


class Workspace:
    """Grow-only scratch buffer per device (partial slabs of the deterministic reductions)."""

    def __init__(self):
        self._buf = {}

    def get(self, nbytes, device):
        nbytes = max(int(nbytes), 256)
        key = str(device)
        buf = self._buf.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=device)
            self._buf[key] = buf
        return buf


_WS = Workspace()


def conv_desc(B, H, W, Cin, Cout, KH, KW, dil):
    return ConvDesc(int(B), int(H), int(W), int(Cin), int(Cout), int(KH), int(KW), int(dil))


def _desc_from(x, w, dil):
    B, H, W, Cin = x.shape
    KH, KW, Ci, Cout = w.shape
    if Ci != Cin:
        raise ValueError("filter Cin %d != activation channels %d" % (Ci, Cin))
    return conv_desc(B, H, W, Cin, Cout, KH, KW, dil)


def conv_fwd(x, w, bias=None, dil=1, relu=True, keep_prob=1.0, drop_mask=None, drop_seed=0, out=None):
    """conv2d/atrous_conv2d SAME stride 1 + bias (+ReLU (+dropout)); deeplab.py:58-59,65-66,71,92-96,100,104."""
    _check(x, name="x"); _check(w, name="w"); _check(bias, name="bias")
    _check(drop_mask, torch.uint8, "drop_mask")
    d = _desc_from(x, w, dil)
    y = out if out is not None else torch.empty((d.B, d.H, d.W, d.Cout), dtype=torch.float32, device=x.device)
    _check(y, name="out")
    epi = EPI_NONE if bias is None else (EPI_BIAS_RELU if relu else EPI_BIAS)
    nbytes = _lib.load().emadapt_conv_fwd_workspace_bytes(ctypes.byref(d))
    ws = _WS.get(nbytes, x.device) if nbytes else None
    _lib.call("emadapt_conv_fwd", ctypes.byref(d), _p(x), _p(w), _p(bias), _p(y), epi, float(keep_prob),
              _p(drop_mask), int(drop_seed) & 0xFFFFFFFFFFFFFFFF, _p(ws), ws.numel() if ws is not None else 0, _stream())
    return y


def conv_dgrad(dy, w, dil=1, gate=None, gate_scale=1.0, out=None):
    """Gradient w.r.t. the conv input, optionally gated by the stored activation that fed the conv."""
    _check(dy, name="dy"); _check(w, name="w"); _check(gate, name="gate")
    B, H, W, Cout = dy.shape
    KH, KW, Cin, Co = w.shape
    if Co != Cout:
        raise ValueError("filter Cout %d != dy channels %d" % (Co, Cout))
    d = conv_desc(B, H, W, Cin, Cout, KH, KW, dil)
    dx = out if out is not None else torch.empty((B, H, W, Cin), dtype=torch.float32, device=dy.device)
    _check(dx, name="out")
    nbytes = _lib.load().emadapt_conv_dgrad_workspace_bytes(ctypes.byref(d))
    ws = _WS.get(nbytes, dy.device) if nbytes else None
    _lib.call("emadapt_conv_dgrad", ctypes.byref(d), _p(dy), _p(w), _p(dx), _p(gate), float(gate_scale), _p(ws),
              ws.numel() if ws is not None else 0, _stream())
    return dx


def conv_wgrad(x, dy, KH, KW, dil=1, out=None, bias_out=None):
    """Gradient w.r.t. the HWIO filter (deterministic split-M reduction).  bias_out (f32 [Cout]): also BiasAddGrad, the column sums
    of dy -- in the same launch where the library fuses it (emadapt_conv_wgrad_bias: the layers on the round-1 kernel), else by
    the separate kernel (bias_grad)."""
    _check(x, name="x"); _check(dy, name="dy"); _check(bias_out, name="bias_out")
    B, H, W, Cin = x.shape
    Cout = dy.shape[3]
    d = conv_desc(B, H, W, Cin, Cout, KH, KW, dil)
    dw = out if out is not None else torch.empty((KH, KW, Cin, Cout), dtype=torch.float32, device=x.device)
    _check(dw, name="out")
    if bias_out is not None:
        if bias_out.numel() != Cout:
            raise ValueError("bias_out has %d elements, the gradient %d channels" % (bias_out.numel(), Cout))
        if _lib.load().emadapt_conv_wgrad_bias_fused(ctypes.byref(d)):
            nbytes = _lib.load().emadapt_conv_wgrad_bias_workspace_bytes(ctypes.byref(d))
            ws = _WS.get(nbytes, x.device)
            _lib.call("emadapt_conv_wgrad_bias", ctypes.byref(d), _p(x), _p(dy), _p(dw), _p(bias_out), _p(ws), ws.numel(), _stream())
            return dw
        bias_grad(dy, out=bias_out)
    nbytes = _lib.load().emadapt_conv_wgrad_workspace_bytes(ctypes.byref(d))
    ws = _WS.get(nbytes, x.device)
    _lib.call("emadapt_conv_wgrad", ctypes.byref(d), _p(x), _p(dy), _p(dw), _p(ws), ws.numel(), _stream())
    return dw


# ---- bf16-compute variants (include/emadapt.h "bf16-compute variants"): f32 tensors in, f32 out; filters come from the
# two bf16 images made by weights_to_bf16 (fwd: [taps, Cout, round8(Cin)], dgrad: [taps, Cin, round8(Cout)]).
def _round8(v):
    return (int(v) + 7) // 8 * 8


def weights_to_bf16(w, fwd_out=None, dgrad_out=None, want_fwd=True, want_dgrad=True):
    """Derive the bf16 filter images from the f32 HWIO master copy; returns (fwd_image, dgrad_image)."""
    _check(w, name="w")
    KH, KW, Cin, Cout = w.shape
    d = conv_desc(1, 1, 1, Cin, Cout, KH, KW, 1)
    wf = wd = None
    if want_fwd:
        wf = fwd_out if fwd_out is not None else torch.empty((KH * KW, Cout, _round8(Cin)), dtype=torch.bfloat16, device=w.device)
        _check(wf, torch.bfloat16, "fwd_out")
        if wf.numel() != KH * KW * Cout * _round8(Cin):
            raise ValueError("fwd_out has the wrong size")
    if want_dgrad:
        wd = dgrad_out if dgrad_out is not None else torch.empty((KH * KW, Cin, _round8(Cout)), dtype=torch.bfloat16, device=w.device)
        _check(wd, torch.bfloat16, "dgrad_out")
        if wd.numel() != KH * KW * Cin * _round8(Cout):
            raise ValueError("dgrad_out has the wrong size")
    _lib.call("emadapt_weights_to_bf16", ctypes.byref(d), _p(w), _p(wf), _p(wd), _stream())
    return wf, wd


W16_MAX_LAYERS = 16


def weights_to_bf16_multi(items):
    """weights_to_bf16 of several filters in one launch: items = [(w, fwd_out or None, dgrad_out or None)], at most W16_MAX_LAYERS;
    the images must exist (sizes as weights_to_bf16 allocates them)."""
    n = len(items)
    if n == 0:
        return
    descs = (ConvDesc * n)()
    ws, wfs, wds = (ctypes.c_void_p * n)(), (ctypes.c_void_p * n)(), (ctypes.c_void_p * n)()
    dps = (ctypes.c_void_p * n)()
    for k, (w, wf, wd) in enumerate(items):
        _check(w, name="w")
        KH, KW, Cin, Cout = w.shape
        if wf is None and wd is None:
            raise ValueError("a layer needs at least one image")
        if wf is not None:
            _check(wf, torch.bfloat16, "fwd_out")
            if wf.numel() != KH * KW * Cout * _round8(Cin):
                raise ValueError("fwd_out has the wrong size")
        if wd is not None:
            _check(wd, torch.bfloat16, "dgrad_out")
            if wd.numel() != KH * KW * Cin * _round8(Cout):
                raise ValueError("dgrad_out has the wrong size")
        descs[k] = conv_desc(1, 1, 1, Cin, Cout, KH, KW, 1)
        dps[k] = ctypes.addressof(descs[k])
        ws[k], wfs[k], wds[k] = w.data_ptr(), (None if wf is None else wf.data_ptr()), (None if wd is None else wd.data_ptr())
    _lib.call("emadapt_weights_to_bf16_multi", n, dps, ws, wfs, wds, _stream())


def conv_fwd_bf16(x, wf, ksize, bias=None, dil=1, relu=True, keep_prob=1.0, drop_mask=None, drop_seed=0, out=None,
                  out_dtype=None):
    """conv_fwd with bf16 operands / fp32 accumulation; wf = forward image of weights_to_bf16, ksize = (KH, KW).
    x may be stored f32 or bf16; the output is f32 unless out / out_dtype say bf16."""
    xdt = _act(x, "x"); _check(wf, torch.bfloat16, "wf"); _check(bias, name="bias")
    _check(drop_mask, torch.uint8, "drop_mask")
    B, H, W, Cin = x.shape
    KH, KW = ksize
    taps, Cout, Cip = wf.shape
    if taps != KH * KW or Cip != _round8(Cin):
        raise ValueError("forward filter image %s does not match x channels %d / ksize %s" % (tuple(wf.shape), Cin, ksize))
    d = conv_desc(B, H, W, Cin, Cout, KH, KW, dil)
    y = out if out is not None else torch.empty((B, H, W, Cout), dtype=out_dtype or torch.float32, device=x.device)
    ydt = _act(y, "out")
    epi = EPI_NONE if bias is None else (EPI_BIAS_RELU if relu else EPI_BIAS)
    nbytes = _lib.load().emadapt_conv_fwd_bf16_workspace_bytes(ctypes.byref(d))
    ws = _WS.get(nbytes, x.device) if nbytes else None
    _lib.call("emadapt_conv_fwd_bf16", ctypes.byref(d), _p(x), xdt, _p(wf), _p(bias), _p(y), ydt, epi, float(keep_prob),
              _p(drop_mask), int(drop_seed) & 0xFFFFFFFFFFFFFFFF, _p(ws), ws.numel() if ws is not None else 0, _stream())
    return y


def conv_dgrad_bf16(dy, wd, ksize, dil=1, gate=None, gate_scale=1.0, out=None, out_dtype=None):
    """conv_dgrad with bf16 operands / fp32 accumulation; wd = dgrad image of weights_to_bf16.
    dy may be stored f32 or bf16; dx (and the gate, the stored activation) are f32 or both bf16."""
    ddt = _act(dy, "dy"); _check(wd, torch.bfloat16, "wd"); gdt = _act(gate, "gate")
    B, H, W, Cout = dy.shape
    KH, KW = ksize
    taps, Cin, Cop = wd.shape
    if taps != KH * KW or Cop != _round8(Cout):
        raise ValueError("dgrad filter image %s does not match dy channels %d / ksize %s" % (tuple(wd.shape), Cout, ksize))
    d = conv_desc(B, H, W, Cin, Cout, KH, KW, dil)
    if out is None:
        out_dtype = out_dtype or (gate.dtype if gate is not None else torch.float32)
    dx = out if out is not None else torch.empty((B, H, W, Cin), dtype=out_dtype, device=dy.device)
    xdt = _act(dx, "out")
    if gate is not None and gdt != xdt:
        raise TypeError("gate and dx must have the same storage type")
    nbytes = _lib.load().emadapt_conv_dgrad_bf16_workspace_bytes(ctypes.byref(d))
    ws = _WS.get(nbytes, dy.device) if nbytes else None
    _lib.call("emadapt_conv_dgrad_bf16", ctypes.byref(d), _p(dy), ddt, _p(wd), _p(dx), xdt, _p(gate), float(gate_scale),
              _p(ws), ws.numel() if ws is not None else 0, _stream())
    return dx


def conv_wgrad_bf16(x, dy, KH, KW, dil=1, out=None, bias_out=None):
    """conv_wgrad with bf16 operands / fp32 accumulation (f32 HWIO result); x and dy stored f32 or bf16.  bias_out (f32 [Cout]):
    also BiasAddGrad -- in the same launch where the library fuses it (dy stored bf16: emadapt_conv_wgrad_bias_bf16), else by the
    separate kernel (bias_grad)."""
    xdt = _act(x, "x"); ddt = _act(dy, "dy")
    B, H, W, Cin = x.shape
    Cout = dy.shape[3]
    d = conv_desc(B, H, W, Cin, Cout, KH, KW, dil)
    dw = out if out is not None else torch.empty((KH, KW, Cin, Cout), dtype=torch.float32, device=x.device)
    _check(dw, name="out")
    if bias_out is not None:
        _check(bias_out, name="bias_out")
        if bias_out.numel() != Cout:
            raise ValueError("bias_out has %d elements, the gradient %d channels" % (bias_out.numel(), Cout))
        if _lib.load().emadapt_conv_wgrad_bias_bf16_fused(ctypes.byref(d), xdt, ddt) and bias_out.data_ptr() % 16 == 0:
            nbytes = _lib.load().emadapt_conv_wgrad_bias_bf16_workspace_bytes(ctypes.byref(d))
            ws = _WS.get(nbytes, x.device)
            _lib.call("emadapt_conv_wgrad_bias_bf16", ctypes.byref(d), _p(x), xdt, _p(dy), ddt, _p(dw), _p(bias_out), _p(ws), ws.numel(),
                      _stream())
            return dw
        bias_grad(dy, out=bias_out)
    nbytes = _lib.load().emadapt_conv_wgrad_bf16_workspace_bytes(ctypes.byref(d))
    ws = _WS.get(nbytes, x.device)
    _lib.call("emadapt_conv_wgrad_bf16", ctypes.byref(d), _p(x), xdt, _p(dy), ddt, _p(dw), _p(ws), ws.numel(), _stream())
    return dw


# ---- bf16 forward / data gradient on LDS-DMA with a tiled filter image (include/emadapt.h "bf16 ... on LDS-DMA")
def conv_bf16t_supported(d, kind, out_bf16=True):
    return bool(_lib.load().emadapt_conv_bf16t_supported(ctypes.byref(d), {"fwd": 0, "dgrad": 1}[kind], 1 if out_bf16 else 0))


def conv_bf16t_pays(d, kind, out_bf16=True):
    """supported AND the faster bf16 kernel for this shape (include/emadapt.h): what the model asks per layer and batch size."""
    return bool(_lib.load().emadapt_conv_bf16t_pays(ctypes.byref(d), {"fwd": 0, "dgrad": 1}[kind], 1 if out_bf16 else 0))


def conv_bf16t_plan(B, H, W, Cin, Cout, k, dil=1, kind="fwd"):
    """(rows of the first launch, 1 = stream-K, its slab levels, slab levels of the launch over the remaining rows / -1)."""
    d = conv_desc(B, H, W, Cin, Cout, k, k, dil)
    out = (ctypes.c_int * 4)()
    _lib.call("emadapt_conv_bf16t_plan", ctypes.byref(d), {"fwd": 0, "dgrad": 1}[kind], out)
    return tuple(out)


def filter_tiles_bf16(w, dgrad=False, out=None):
    """Tiled bf16 image of an f32 HWIO filter for conv_fwd_bf16t (dgrad=False) / conv_dgrad_bf16t (dgrad=True)."""
    _check(w, name="w")
    KH, KW, Cin, Cout = w.shape
    d = conv_desc(1, 1, 1, Cin, Cout, KH, KW, 1)
    n = _lib.load().emadapt_filter_tiles_bf16_bytes(ctypes.byref(d), int(bool(dgrad)))
    if n == 0:
        raise ValueError("no tiled bf16 image for a filter with %d reduction channels (multiple of 32 needed)" % (Cout if dgrad else Cin))
    t = out if out is not None else torch.empty(n // 2, dtype=torch.bfloat16, device=w.device)
    _check(t, torch.bfloat16, "out")
    if t.numel() * 2 != n:
        raise ValueError("out has the wrong size")
    _lib.call("emadapt_filter_tiles_bf16", ctypes.byref(d), _p(w), _p(t), int(bool(dgrad)), _stream())
    return t


def filter_tiles_bf16_both(w, out_fwd, out_dgrad):
    """Both tiled bf16 images of an f32 HWIO filter in one pass (the model's refresh where a layer's forward AND data gradient run on the
    LDS-DMA kernel)."""
    _check(w, name="w")
    KH, KW, Cin, Cout = w.shape
    d = conv_desc(1, 1, 1, Cin, Cout, KH, KW, 1)
    for t, dg in ((out_fwd, 0), (out_dgrad, 1)):
        n = _lib.load().emadapt_filter_tiles_bf16_bytes(ctypes.byref(d), dg)
        if n == 0 or t.dtype != torch.bfloat16 or t.numel() * 2 != n:
            raise ValueError("tiled image buffer does not match the filter")
    _lib.call("emadapt_filter_tiles_bf16_both", ctypes.byref(d), _p(w), _p(out_fwd), _p(out_dgrad), _stream())
    return out_fwd, out_dgrad


def conv_fwd_bf16t(x, wt, ksize, Cout, bias=None, dil=1, relu=True, keep_prob=1.0, drop_mask=None, drop_seed=0, out=None,
                   out_dtype=None):
    """conv_fwd_bf16 on the LDS-DMA kernel: x stored bf16, wt = filter_tiles_bf16(w); output bf16 unless out / out_dtype say f32."""
    _check(x, torch.bfloat16, "x"); _check(wt, torch.bfloat16, "wt"); _check(bias, name="bias"); _check(drop_mask, torch.uint8, "drop_mask")
    B, H, W, Cin = x.shape
    KH, KW = ksize
    d = conv_desc(B, H, W, Cin, Cout, KH, KW, dil)
    if wt.numel() * 2 != _lib.load().emadapt_filter_tiles_bf16_bytes(ctypes.byref(d), 0):
        raise ValueError("tiled filter image does not match the convolution")
    y = out if out is not None else torch.empty((B, H, W, Cout), dtype=out_dtype or torch.bfloat16, device=x.device)
    ydt = _act(y, "out")
    epi = EPI_NONE if bias is None else (EPI_BIAS_RELU if relu else EPI_BIAS)
    nbytes = _lib.load().emadapt_conv_bf16t_workspace_bytes(ctypes.byref(d), 0)
    ws = _WS.get(nbytes, x.device) if nbytes else None
    _lib.call("emadapt_conv_fwd_bf16t", ctypes.byref(d), _p(x), _p(wt), _p(bias), _p(y), ydt, epi, float(keep_prob), _p(drop_mask),
              int(drop_seed) & 0xFFFFFFFFFFFFFFFF, _p(ws), ws.numel() if ws is not None else 0, _stream())
    return y


def conv_dgrad_bf16t(dy, wt, ksize, Cin, dil=1, gate=None, gate_scale=1.0, out=None, out_dtype=None):
    """conv_dgrad_bf16 on the LDS-DMA kernel: dy stored bf16, wt = filter_tiles_bf16(w, dgrad=True); dx and the gate share a type."""
    _check(dy, torch.bfloat16, "dy"); _check(wt, torch.bfloat16, "wt"); gdt = _act(gate, "gate")
    B, H, W, Cout = dy.shape
    KH, KW = ksize
    d = conv_desc(B, H, W, Cin, Cout, KH, KW, dil)
    if wt.numel() * 2 != _lib.load().emadapt_filter_tiles_bf16_bytes(ctypes.byref(d), 1):
        raise ValueError("tiled filter image does not match the convolution")
    if out is None:
        out_dtype = out_dtype or (gate.dtype if gate is not None else torch.bfloat16)
    dx = out if out is not None else torch.empty((B, H, W, Cin), dtype=out_dtype, device=dy.device)
    xdt = _act(dx, "out")
    if gate is not None and gdt != xdt:
        raise TypeError("gate and dx must have the same storage type")
    nbytes = _lib.load().emadapt_conv_bf16t_workspace_bytes(ctypes.byref(d), 1)
    ws = _WS.get(nbytes, dy.device) if nbytes else None
    _lib.call("emadapt_conv_dgrad_bf16t", ctypes.byref(d), _p(dy), _p(wt), _p(dx), xdt, _p(gate), float(gate_scale),
              _p(ws), ws.numel() if ws is not None else 0, _stream())
    return dx


# ---- fp32-grade GEMMs on the bf16 matrix pipe (include/emadapt.h "bf16x6"): all tensors f32
def filter_kmajor(w, out=None):
    """k-contiguous f32 image [KH*KW, Cout, Cin] of the HWIO filter (what conv_fwd_x6 reads)."""
    _check(w, name="w")
    KH, KW, Cin, Cout = w.shape
    d = conv_desc(1, 1, 1, Cin, Cout, KH, KW, 1)
    wk = out if out is not None else torch.empty((KH * KW, Cout, Cin), dtype=torch.float32, device=w.device)
    _check(wk, name="out")
    if wk.numel() != w.numel():
        raise ValueError("out has the wrong size")
    _lib.call("emadapt_filter_kmajor", ctypes.byref(d), _p(w), _p(wk), _stream())
    return wk


def conv_fwd_x6(x, wk, ksize, bias=None, dil=1, relu=True, keep_prob=1.0, drop_mask=None, drop_seed=0, out=None):
    """conv_fwd computed as six bf16 MFMAs per product over the exact 3-way bf16 split of both operands."""
    _check(x, name="x"); _check(wk, name="wk"); _check(bias, name="bias"); _check(drop_mask, torch.uint8, "drop_mask")
    B, H, W, Cin = x.shape
    KH, KW = ksize
    taps, Cout, Ci = wk.shape
    if taps != KH * KW or Ci != Cin:
        raise ValueError("k-major filter image %s does not match x channels %d / ksize %s" % (tuple(wk.shape), Cin, ksize))
    d = conv_desc(B, H, W, Cin, Cout, KH, KW, dil)
    y = out if out is not None else torch.empty((B, H, W, Cout), dtype=torch.float32, device=x.device)
    _check(y, name="out")
    epi = EPI_NONE if bias is None else (EPI_BIAS_RELU if relu else EPI_BIAS)
    nbytes = _lib.load().emadapt_conv_fwd_x6_workspace_bytes(ctypes.byref(d))
    ws = _WS.get(nbytes, x.device) if nbytes else None
    _lib.call("emadapt_conv_fwd_x6", ctypes.byref(d), _p(x), _p(wk), _p(bias), _p(y), epi, float(keep_prob),
              _p(drop_mask), int(drop_seed) & 0xFFFFFFFFFFFFFFFF, _p(ws), ws.numel() if ws is not None else 0, _stream())
    return y


def conv_dgrad_x6(dy, w, dil=1, gate=None, gate_scale=1.0, out=None):
    _check(dy, name="dy"); _check(w, name="w"); _check(gate, name="gate")
    B, H, W, Cout = dy.shape
    KH, KW, Cin, Co = w.shape
    if Co != Cout:
        raise ValueError("filter Cout %d != dy channels %d" % (Co, Cout))
    d = conv_desc(B, H, W, Cin, Cout, KH, KW, dil)
    dx = out if out is not None else torch.empty((B, H, W, Cin), dtype=torch.float32, device=dy.device)
    _check(dx, name="out")
    nbytes = _lib.load().emadapt_conv_dgrad_x6_workspace_bytes(ctypes.byref(d))
    ws = _WS.get(nbytes, dy.device) if nbytes else None
    _lib.call("emadapt_conv_dgrad_x6", ctypes.byref(d), _p(dy), _p(w), _p(dx), _p(gate), float(gate_scale), _p(ws),
              ws.numel() if ws is not None else 0, _stream())
    return dx


def conv_wgrad_x6(x, dy, KH, KW, dil=1, out=None):
    _check(x, name="x"); _check(dy, name="dy")
    B, H, W, Cin = x.shape
    Cout = dy.shape[3]
    d = conv_desc(B, H, W, Cin, Cout, KH, KW, dil)
    dw = out if out is not None else torch.empty((KH, KW, Cin, Cout), dtype=torch.float32, device=x.device)
    _check(dw, name="out")
    nbytes = _lib.load().emadapt_conv_wgrad_x6_workspace_bytes(ctypes.byref(d))
    ws = _WS.get(nbytes, x.device)
    _lib.call("emadapt_conv_wgrad_x6", ctypes.byref(d), _p(x), _p(dy), _p(dw), _p(ws), ws.numel(), _stream())
    return dw


# ---- bf16x6 on pre-split operands (csrc/conv_x6p.hip): plane images [rows][C/16][3][16] bf16, stored as uint8 tensors
def split_planes(x, out=None):
    """Plane image of an f32 tensor [..., C] (C % 16 == 0): x = p0 + p1 + p2 exactly, three bf16 planes, 6 bytes per element."""
    _check(x, name="x")
    C = x.shape[-1]
    rows = x.numel() // C
    nbytes = rows * C * 6
    planes = out if out is not None else torch.empty((nbytes,), dtype=torch.uint8, device=x.device)
    _check(planes, torch.uint8, "out")
    if planes.numel() < nbytes:
        raise ValueError("out is too small for the plane image")
    _lib.call("emadapt_split_planes", _p(x), _p(planes), rows, C, _stream())
    return planes


def filter_tiles(w, dgrad=False, out=None):
    """Tiled plane image of an HWIO filter for the plane-image forward (rows = Cout, k = Cin) or data-gradient (rows = Cin,
    k = Cout) kernel: [tap][k / 16][rows padded to 256][96 B in LDS order] (csrc/igemm_x6p.h)."""
    _check(w, name="w")
    KH, KW, Cin, Cout = w.shape
    d = conv_desc(1, 1, 1, Cin, Cout, KH, KW, 1)
    nbytes = _lib.load().emadapt_filter_tiles_bytes(ctypes.byref(d), int(bool(dgrad)))
    if nbytes == 0:
        raise ValueError("the reduction's channel count must be a whole number of groups of 16")
    tiles = out if out is not None else torch.empty((nbytes,), dtype=torch.uint8, device=w.device)
    _check(tiles, torch.uint8, "out")
    if tiles.numel() < nbytes:
        raise ValueError("out is too small for the tiled filter image")
    _lib.call("emadapt_filter_tiles", ctypes.byref(d), _p(w), _p(tiles), int(bool(dgrad)), _stream())
    return tiles


def filter_tiles_bytes(Cin, Cout, KH, KW, dgrad=False):
    d = conv_desc(1, 1, 1, Cin, Cout, KH, KW, 1)
    return int(_lib.load().emadapt_filter_tiles_bytes(ctypes.byref(d), int(bool(dgrad))))


def conv_x6p_supported(d, kind):
    """kind 0 forward, 1 data gradient, 2 filter gradient: does the plane-image kernel take this convolution?"""
    return bool(_lib.load().emadapt_conv_x6p_supported(ctypes.byref(d), int(kind)))


def conv_x6p_pays(d, kind):
    """conv_x6p_supported and enough work for one 256 x 256 workgroup per CU (a host with both kernel families asks this)."""
    return bool(_lib.load().emadapt_conv_x6p_pays(ctypes.byref(d), int(kind)))


def conv_x6p_plan(d, dgrad=False):
    plan = (ctypes.c_int * 5)()
    _lib.call("emadapt_conv_x6p_plan", ctypes.byref(d), int(bool(dgrad)), plan)
    return tuple(plan)


def conv_fwd_x6p(xp, wkp, d, bias=None, relu=True, keep_prob=1.0, drop_mask=None, drop_seed=0, out=None):
    """Forward convolution d from the plane image of x and the tiled filter image filter_tiles(w)."""
    _check(xp, torch.uint8, "xp"); _check(wkp, torch.uint8, "wkp"); _check(bias, name="bias"); _check(drop_mask, torch.uint8, "drop_mask")
    y = out if out is not None else torch.empty((d.B, d.H, d.W, d.Cout), dtype=torch.float32, device=xp.device)
    _check(y, name="out")
    epi = EPI_NONE if bias is None else (EPI_BIAS_RELU if relu else EPI_BIAS)
    nbytes = _lib.load().emadapt_conv_fwd_x6p_workspace_bytes(ctypes.byref(d))
    ws = _WS.get(nbytes, xp.device) if nbytes else None
    _lib.call("emadapt_conv_fwd_x6p", ctypes.byref(d), _p(xp), _p(wkp), _p(bias), _p(y), epi, float(keep_prob),
              _p(drop_mask), int(drop_seed) & 0xFFFFFFFFFFFFFFFF, _p(ws), ws.numel() if ws is not None else 0, _stream())
    return y


def conv_dgrad_x6p(dyp, wp, d, gate=None, gate_scale=1.0, out=None):
    """Data gradient of convolution d from the plane image of dy and the tiled filter image filter_tiles(w, dgrad=True)."""
    _check(dyp, torch.uint8, "dyp"); _check(wp, torch.uint8, "wp"); _check(gate, name="gate")
    dx = out if out is not None else torch.empty((d.B, d.H, d.W, d.Cin), dtype=torch.float32, device=dyp.device)
    _check(dx, name="out")
    nbytes = _lib.load().emadapt_conv_dgrad_x6p_workspace_bytes(ctypes.byref(d))
    ws = _WS.get(nbytes, dyp.device) if nbytes else None
    _lib.call("emadapt_conv_dgrad_x6p", ctypes.byref(d), _p(dyp), _p(wp), _p(dx), _p(gate), float(gate_scale), _p(ws),
              ws.numel() if ws is not None else 0, _stream())
    return dx


def conv_wgrad_x6p(xp, dyp, d, out=None):
    """Filter gradient of convolution d from the plane images of x and dy."""
    _check(xp, torch.uint8, "xp"); _check(dyp, torch.uint8, "dyp")
    dw = out if out is not None else torch.empty((d.KH, d.KW, d.Cin, d.Cout), dtype=torch.float32, device=xp.device)
    _check(dw, name="out")
    nbytes = _lib.load().emadapt_conv_wgrad_x6p_workspace_bytes(ctypes.byref(d))
    ws = _WS.get(nbytes, xp.device)
    _lib.call("emadapt_conv_wgrad_x6p", ctypes.byref(d), _p(xp), _p(dyp), _p(dw), _p(ws), ws.numel(), _stream())
    return dw


def bias_grad(dy, out=None):
    dt = _act(dy, "dy")
    N = dy.shape[-1]
    M = dy.numel() // N
    db = out if out is not None else torch.empty((N,), dtype=torch.float32, device=dy.device)
    nbytes = _lib.load().emadapt_bias_grad_workspace_bytes(M, N)
    ws = _WS.get(nbytes, dy.device)
    if dt == DT_F32:
        _lib.call("emadapt_bias_grad", _p(dy), M, N, _p(db), _p(ws), ws.numel(), _stream())
    else:
        _lib.call("emadapt_bias_grad_dt", _p(dy), dt, M, N, _p(db), _p(ws), ws.numel(), _stream())
    return db


def pool_out_size(n, stride):
    return (n + stride - 1) // stride


POOL_CLOSED = 0x10   # bit of a gated max-pool index: the window's winner is not > 0 (include/emadapt.h emadapt_maxpool_fwd_gated)


def maxpool_fwd(x, stride, out=None, idx=None, relu_gate=False):
    """3x3 max-pool, SAME (deeplab.py:77,82). Returns (y, idx) with idx the winning tap (uint8).  relu_gate: the index also carries
    the ReluGrad gate of x (POOL_CLOSED), and maxpool_bwd on it needs no gate tensor."""
    dt = _act(x, "x")
    B, H, W, C = x.shape
    OH, OW = pool_out_size(H, stride), pool_out_size(W, stride)
    y = out if out is not None else torch.empty((B, OH, OW, C), dtype=x.dtype, device=x.device)
    if _act(y, "out") != dt:
        raise TypeError("max-pool input and output must have the same storage type")
    if idx is None:
        idx = torch.empty((B, OH, OW, C), dtype=torch.uint8, device=x.device)
    if relu_gate:
        _lib.call("emadapt_maxpool_fwd_gated", _p(x), dt, B, H, W, C, int(stride), _p(y), _p(idx), 1, _stream())
    elif dt == DT_F32:
        _lib.call("emadapt_maxpool_fwd", _p(x), B, H, W, C, int(stride), _p(y), _p(idx), _stream())
    else:
        _lib.call("emadapt_maxpool_fwd_dt", _p(x), dt, B, H, W, C, int(stride), _p(y), _p(idx), _stream())
    return y, idx


def maxpool_bwd(dy, idx, in_shape, stride, gate=None, out=None):
    dt = _act(dy, "dy"); _check(idx, torch.uint8, "idx"); _act(gate, "gate")
    B, H, W, C = in_shape
    dx = out if out is not None else torch.empty((B, H, W, C), dtype=dy.dtype, device=dy.device)
    if _act(dx, "out") != dt or (gate is not None and gate.dtype != dy.dtype):
        raise TypeError("max-pool gradient, gate and output must have the same storage type")
    if dt == DT_F32:
        _lib.call("emadapt_maxpool_bwd", _p(dy), _p(idx), _p(gate), B, H, W, C, int(stride), _p(dx), _stream())
    else:
        _lib.call("emadapt_maxpool_bwd_dt", _p(dy), _p(idx), _p(gate), dt, B, H, W, C, int(stride), _p(dx), _stream())
    return dx


def copy_pad_channels(src, Cdst, out=None):
    _check(src, name="src")
    Cs = src.shape[-1]
    rows = src.numel() // Cs
    dst = out if out is not None else torch.empty(tuple(src.shape[:-1]) + (Cdst,), dtype=torch.float32,
                                                  device=src.device)
    _lib.call("emadapt_copy_pad_channels", _p(src), rows, Cs, _p(dst), int(Cdst), _stream())
    return dst


def resize_nearest(label, oh, ow, out=None):
    """tf.image.resize_nearest_neighbor + squeeze (deeplab.py:110). label: f32 [B,H,W] or [B,H,W,1]."""
    _check(label, name="label")
    if label.dim() == 4:
        label = label.reshape(label.shape[:3])
    B, H, W = label.shape
    y = out if out is not None else torch.empty((B, oh, ow), dtype=torch.float32, device=label.device)
    _lib.call("emadapt_resize_nearest", _p(label), B, H, W, _p(y), int(oh), int(ow), _stream())
    return y


def pred_bilinear_argmax(logits, num_classes, out_h, out_w, want_rescale=False):
    """network.py:39-41: resize_bilinear to (out_h, out_w) + argmax over classes. logits f32 [B,h,w,ldl]."""
    _check(logits, name="logits")
    B, IH, IW, ldl = logits.shape
    pred = torch.empty((B, out_h, out_w), dtype=torch.int64, device=logits.device)
    resc = torch.empty((B, out_h, out_w, int(num_classes)), dtype=torch.float32, device=logits.device) if want_rescale else None
    _lib.call("emadapt_pred_bilinear_argmax", _p(logits), ldl, B, IH, IW, int(num_classes), int(out_h), int(out_w),
              _p(pred), _p(resc), _stream())
    return pred, resc


def resize_bilinear(x, out_h, out_w):
    """tf.image.resize_bilinear (TF1 legacy mapping, align_corners=False). x f32 [B,H,W,C]."""
    _check(x, name="x")
    B, IH, IW, C = x.shape
    out = torch.empty((B, out_h, out_w, C), dtype=torch.float32, device=x.device)
    _lib.call("emadapt_pred_bilinear_argmax", _p(x), C, B, IH, IW, C, int(out_h), int(out_w), None, _p(out), _stream())
    return out


IMG_MEAN_BGR = (104.00698793, 116.66876762, 122.67891434)  # dataset.py:16-18


def bgr_mean_flip(img_rgb, label=None, flip=None, mean_bgr=IMG_MEAN_BGR):
    """dataset.py:175-181 + image_flip: RGB->BGR, minus mean, optional per-image left-right flip (uint8 [B])."""
    _check(img_rgb, name="img_rgb"); _check(label, name="label"); _check(flip, torch.uint8, "flip")
    B, H, W, _ = img_rgb.shape
    out = torch.empty_like(img_rgb)
    lab_out = torch.empty_like(label) if label is not None else None
    mean = (ctypes.c_float * 3)(*[float(m) for m in mean_bgr])
    _lib.call("emadapt_bgr_mean_flip", _p(img_rgb), _p(label), B, H, W, mean, _p(flip), _p(out), _p(lab_out), _stream())
    return out, lab_out


def voc_preprocess(staging, samples, out_h, out_w, img_out=None, label_out=None, mean_bgr=IMG_MEAN_BGR):
    """dataset.image_preprocess (dataset.py:147-185) of a whole batch in one launch (include/emadapt.h, emadapt_voc_preprocess).
    staging: uint8 device tensor holding the decoded samples; samples: list of (img_off, lab_off, h0, w0, h1, w1, flip) -- h1 = w1 = 0
    without the random-scale stage.  Returns (images f32 [B,H,W,3] BGR - mean, labels f32 [B,H,W,1])."""
    _check(staging, torch.uint8, "staging")
    B = len(samples)
    arr = (_lib.VocSample * max(B, 1))()
    for i, t in enumerate(samples):
        arr[i].img_off, arr[i].lab_off, arr[i].h0, arr[i].w0, arr[i].h1, arr[i].w1, arr[i].flip = [int(v) for v in t]
    if img_out is None:
        img_out = torch.empty((B, out_h, out_w, 3), dtype=torch.float32, device=staging.device)
    if label_out is None:
        label_out = torch.empty((B, out_h, out_w, 1), dtype=torch.float32, device=staging.device)
    _check(img_out, name="img_out"); _check(label_out, name="label_out")
    if img_out.numel() != B * out_h * out_w * 3 or label_out.numel() != B * out_h * out_w:
        raise ValueError("img_out / label_out must hold B x H x W (x 3) elements")
    mean = (ctypes.c_float * 3)(*[float(m) for m in mean_bgr])
    _lib.call("emadapt_voc_preprocess", _p(staging), staging.numel(), arr, B, int(out_h), int(out_w), mean, _p(img_out), _p(label_out), _stream())
    return img_out, label_out


def estep_device(logits, label, perms, num_classes, suppress_others=True, num_iter=5, margin_others=1e-5,
                 bg_p=0.5, fg_p=0.25, label_kind=LABEL_F32, want_map=False, labels_out=None, map_out=None):
    """Device E-step (deeplab.py:109-124 without the host round trip).

    logits: f32 [B,H,W,ldl] with ldl >= num_classes (padded class dimension allowed);
    label : f32/i32 pixel labels [B, ...] or uint8 multi-hot [B,C]; perms: int32 [num_iter, C-1] (device).
    Returns (labels int64 [B,H,W], float_map or None).
    """
    _check(logits, name="logits")
    B, H, W, ldl = logits.shape
    C = int(num_classes)
    dt = {LABEL_F32: torch.float32, LABEL_I32: torch.int32, LABEL_MULTIHOT: torch.uint8}[label_kind]
    _check(label, dt, "label")
    _check(perms, torch.int32, "perms")
    lib = _lib.load()
    label_len = 0 if label_kind == LABEL_MULTIHOT else (label.numel() // B if B else 0)
    if labels_out is None:
        labels_out = torch.empty((B, H, W), dtype=torch.int64, device=logits.device)
    hw = H * W
    k_bg, k_fg = int(hw * bg_p), int(hw * fg_p)  # estep.py:58-59
    if lib.emadapt_estep_needs_map(H, W, C) < 0:
        raise ValueError("E-step geometry %dx%dx%d is not supported (H*W <= 8192, C <= 256)" % (H, W, C))
    if not want_map:
        map_out = None   # only an output: the channels that do not fit in LDS live in the workspace
    elif map_out is None:
        map_out = torch.empty((B, H, W, C), dtype=torch.float32, device=logits.device)
    nbytes = lib.emadapt_estep_workspace_bytes(B, H, W, C)
    ws = _WS.get(nbytes + 256, logits.device)
    _lib.call("emadapt_estep", _p(logits), ldl, _p(label), label_kind, label_len, _p(perms), B, H, W, C,
              int(bool(suppress_others)), int(num_iter), float(margin_others), k_bg, k_fg, _p(map_out),
              _p(labels_out), _p(ws), ws.numel(), _stream())
    return labels_out, map_out


def softmax_ce(logits, labels, num_classes, grad_scale=None, want_grad=True, want_loss=True, dlogits=None):
    """sparse softmax cross-entropy (sum over pixels) and its gradient (deeplab.py:182).

    logits f32 [..., ldl]; labels int64 [...]. Returns (loss_sum f32[1] or None, dlogits or None)."""
    _check(logits, name="logits"); _check(labels, torch.int64, "labels")
    ldl = logits.shape[-1]
    M = logits.numel() // ldl
    if grad_scale is None:
        grad_scale = 1.0 / M
    if want_grad and dlogits is None:
        dlogits = torch.empty_like(logits)
    ddt = _act(dlogits, "dlogits") if want_grad else DT_F32
    loss = torch.empty((1,), dtype=torch.float32, device=logits.device) if want_loss else None
    nbytes = _lib.load().emadapt_softmax_ce_workspace_bytes(M)
    ws = _WS.get(nbytes, logits.device)
    if ddt == DT_F32:
        _lib.call("emadapt_softmax_ce", _p(logits), ldl, _p(labels), M, int(num_classes), float(grad_scale),
                  _p(dlogits) if want_grad else None, _p(loss), _p(ws), ws.numel(), _stream())
    else:  # gradient stored bf16 (consumed by the bf16-compute fc8 kernels)
        _lib.call("emadapt_softmax_ce_dt", _p(logits), ldl, _p(labels), M, int(num_classes), float(grad_scale),
                  _p(dlogits), ddt, _p(loss), _p(ws), ws.numel(), _stream())
    return loss, (dlogits if want_grad else None)


def l2_loss(x):
    _check(x, name="x")
    out = torch.empty((1,), dtype=torch.float32, device=x.device)
    ws = _WS.get(_lib.load().emadapt_l2_loss_workspace_bytes(x.numel()), x.device)
    _lib.call("emadapt_l2_loss", _p(x), x.numel(), _p(out), _p(ws), ws.numel(), _stream())
    return out


def sgd_accumulate(accum, g, w, n_decay, weight_decay, scale):
    _check(accum, name="accum"); _check(g, name="g"); _check(w, name="w")
    _lib.call("emadapt_sgd_accumulate", _p(accum), _p(g), _p(w), accum.numel(), int(n_decay), float(weight_decay),
              float(scale), _stream())


def sgd_momentum(w, v, accum, lr, momentum, accum_scale=1.0):
    _check(w, name="w"); _check(v, name="v"); _check(accum, name="accum")
    _lib.call("emadapt_sgd_momentum", _p(w), _p(v), _p(accum), w.numel(), float(lr), float(momentum),
              float(accum_scale), _stream())


def sgd_accumulate_momentum(w, v, g, n_decay, weight_decay, gscale, lr, momentum, ascale=1.0):
    """accumulate + Momentum + clean of one micro-batch in one pass (accum_num == 1, no exchange): bit-identical weights."""
    _check(w, name="w"); _check(v, name="v"); _check(g, name="g")
    _lib.call("emadapt_sgd_accumulate_momentum", _p(w), _p(v), _p(g), w.numel(), int(n_decay), float(weight_decay), float(gscale),
              float(lr), float(momentum), float(ascale), _stream())


SGD_MAX_SEGS = 16


def sgd_accumulate_momentum_multi(w, v, g, segs, weight_decay, gscale, lr, momentum, ascale=1.0):
    """sgd_accumulate_momentum over several segments (start, n, n_decay) of the flat buffers in one launch (at most SGD_MAX_SEGS,
    multiples of 4 each): bit-identical to one call per segment."""
    _check(w, name="w"); _check(v, name="v"); _check(g, name="g")
    k = len(segs)
    arr = ctypes.c_int64 * max(k, 1)
    st, nn, nd = arr(*[int(s[0]) for s in segs]), arr(*[int(s[1]) for s in segs]), arr(*[int(s[2]) for s in segs])
    if any(s[0] + s[1] > w.numel() for s in segs) or v.numel() != w.numel() or g.numel() != w.numel():
        raise ValueError("segment outside the buffers")
    _lib.call("emadapt_sgd_accumulate_momentum_multi", _p(w), _p(v), _p(g), k, st, nn, nd, float(weight_decay), float(gscale), float(lr),
              float(momentum), float(ascale), _stream())


def sgd_accumulate_momentum_tiles_supported(KH, KW, Cin, Cout):
    d = conv_desc(1, 1, 1, Cin, Cout, KH, KW, 1)
    return bool(_lib.load().emadapt_sgd_accumulate_momentum_tiles_supported(ctypes.byref(d)))


def sgd_accumulate_momentum_tiles(w, v, g, filter_shape, weight_decay, gscale, lr, momentum, ascale, out_fwd, out_dgrad):
    """sgd_accumulate_momentum on one layer's [filter | biases] segment that also writes both tiled bf16 images of the new filter
    (filter_tiles_bf16_both): bit-identical weights, momentum and images, without the refresh pass's second read of the weights."""
    _check(w, name="w"); _check(v, name="v"); _check(g, name="g")
    KH, KW, Cin, Cout = filter_shape
    nw = KH * KW * Cin * Cout
    if w.numel() < nw or v.numel() != w.numel() or g.numel() != w.numel():
        raise ValueError("w / v / g must hold the %d filter elements followed by the biases" % nw)
    d = conv_desc(1, 1, 1, Cin, Cout, KH, KW, 1)
    for t, dg in ((out_fwd, 0), (out_dgrad, 1)):
        n = _lib.load().emadapt_filter_tiles_bf16_bytes(ctypes.byref(d), dg)
        if n == 0 or t.dtype != torch.bfloat16 or t.numel() * 2 != n or not t.is_cuda or not t.is_contiguous():
            raise ValueError("tiled image buffer does not match the filter")
    _lib.call("emadapt_sgd_accumulate_momentum_tiles", ctypes.byref(d), _p(w), _p(v), _p(g), w.numel() - nw, float(weight_decay),
              float(gscale), float(lr), float(momentum), float(ascale), _p(out_fwd), _p(out_dgrad), _stream())


def sgd_momentum_payload_tiles(w, v, accum, payload, filter_shape, lr, momentum, scale, out_fwd, out_dgrad):
    """sgd_momentum_payload on one layer's [filter | biases] segment that also writes both tiled bf16 images of the new filter."""
    _check(w, name="w"); _check(v, name="v"); _check(accum, name="accum")
    pdt = _payload_dt(payload, "payload")
    KH, KW, Cin, Cout = filter_shape
    nw = KH * KW * Cin * Cout
    if w.numel() < nw or v.numel() != w.numel() or payload.numel() != w.numel() or (accum is not None and accum.numel() != w.numel()):
        raise ValueError("w / v / accum / payload must hold the %d filter elements followed by the biases" % nw)
    d = conv_desc(1, 1, 1, Cin, Cout, KH, KW, 1)
    for t, dg in ((out_fwd, 0), (out_dgrad, 1)):
        n = _lib.load().emadapt_filter_tiles_bf16_bytes(ctypes.byref(d), dg)
        if n == 0 or t.dtype != torch.bfloat16 or t.numel() * 2 != n or not t.is_cuda or not t.is_contiguous():
            raise ValueError("tiled image buffer does not match the filter")
    _lib.call("emadapt_sgd_momentum_payload_tiles", ctypes.byref(d), _p(w), _p(v), _p(accum), _p(payload), pdt, w.numel() - nw, float(lr),
              float(momentum), float(scale), _p(out_fwd), _p(out_dgrad), _stream())


def _payload_dt(t, name):
    if not t.is_cuda or not t.is_contiguous():
        raise ValueError("%s must be a contiguous GPU tensor" % name)
    if t.dtype == torch.float32:
        return DT_F32
    if t.dtype == torch.bfloat16:
        return DT_BF16
    raise TypeError("%s must be float32 or bfloat16, got %s" % (name, t.dtype))


def sgd_accumulate_pack(g, w, n_decay, weight_decay, scale, out):
    """out[i] = payload type of 0 + (g[i] + weight_decay * w[i] (i < n_decay)) * scale: accumulate + pack without an accumulator (accum_num == 1,
    data-parallel exchange fed `prepacked`: dist.py)."""
    _check(g, name="g"); _check(w, name="w")
    dt = _payload_dt(out, "out")
    if out.numel() != g.numel() or w.numel() != g.numel():
        raise ValueError("g, w and out must have the same length")
    _lib.call("emadapt_sgd_accumulate_pack", _p(g), _p(w), g.numel(), int(n_decay), float(weight_decay), float(scale), _p(out), dt, _stream())
    return out


def sgd_momentum_payload(w, v, accum, payload, lr, momentum, scale=1.0):
    """Momentum reading the summed gradient from the gathered payload of the gradient exchange (f32 or bf16); accum is zeroed (None: there is
    no accumulator to zero)."""
    _check(w, name="w"); _check(v, name="v")
    if accum is not None:
        _check(accum, name="accum")
    dt = _payload_dt(payload, "payload")
    if payload.numel() < w.numel():
        raise ValueError("payload holds %d elements, the parameters %d" % (payload.numel(), w.numel()))
    _lib.call("emadapt_sgd_momentum_payload", _p(w), _p(v), _p(accum), _p(payload), dt, w.numel(), float(lr), float(momentum),
              float(scale), _stream())


def grad_pack(src, dst):
    """dst[:n] = payload(src) (bf16: round to nearest even), dst[n:] = 0.  src f32 [n]; dst f32 or bf16 [npad >= n]."""
    _check(src, name="src")
    dt = _payload_dt(dst, "dst")
    if dst.numel() < src.numel():
        raise ValueError("dst is shorter than src")
    _lib.call("emadapt_grad_pack", _p(src), src.numel(), _p(dst), dt, dst.numel(), _stream())
    return dst


def grad_owner_sum(recv, world, out):
    """out[i] = payload(sum_r f32(recv[r * shard + i])), fp32 adds in rank order; recv [world * shard], out [shard]."""
    dt = _payload_dt(recv, "recv")
    if _payload_dt(out, "out") != dt or recv.numel() != world * out.numel():
        raise ValueError("recv must be [world x shard] of out's type")
    _lib.call("emadapt_grad_owner_sum", _p(recv), dt, int(world), out.numel(), _p(out), _stream())
    return out


def status_word():
    """Device address of the status word (tests)."""
    p = ctypes.c_void_p()
    _lib.call("emadapt_status_word", ctypes.byref(p))
    return p.value


def status_poll(clear=True, raise_on_error=True):
    """Copy the device status word to the host on the current stream and SYNCHRONISE it; returns the flags (0 = clean).  With
    raise_on_error a raised word becomes EmadaptDeviceError (after clearing it when `clear`)."""
    flags = ctypes.c_uint(0)
    rc = _lib.load().emadapt_status_poll(ctypes.byref(flags), 1 if clear else 0, _stream())
    if rc not in (0, -6):
        raise _lib.EmadaptError(rc, "emadapt_status_poll")
    if flags.value and raise_on_error:
        raise _lib.EmadaptDeviceError(flags.value, "device status 0x%x" % flags.value)
    return flags.value


def status_collective(on):
    """Collective mode of the status word for data-parallel jobs (include/emadapt.h, emadapt_status_collective): the failure travels
    with the exchanged gradient (NaN) instead of being decided from each rank's own word.  Returns the previous setting."""
    return bool(_lib.load().emadapt_status_collective(1 if on else 0))


def status_raise(bits):
    """Raise bits of the device status word from the host, in stream order (include/emadapt.h, emadapt_status_raise)."""
    _lib.call("emadapt_status_raise", int(bits), _stream())


STATUS_STREAMK_TIMEOUT = _lib.STATUS_STREAMK_TIMEOUT
STATUS_NAN_GRADIENT = _lib.STATUS_NAN_GRADIENT


def conv_plan(B, H, W, Cin, Cout, k, dil=1, dgrad=False):
    """(tile variant, main rows, main split-K factor / for stream-K the most workgroups sharing a tile, tail split-K -- or, for a stream-K launch that skips
    the zero padding of a dilated filter, its number of pixel classes) of the fp32 forward / data-gradient launch plan."""
    d = conv_desc(B, H, W, Cin, Cout, k, k, dil)
    out = (ctypes.c_int * 4)()
    _lib.call("emadapt_conv_plan", ctypes.byref(d), 1 if dgrad else 0, out)
    return tuple(out)


def conv_bf16_plan(B, H, W, Cin, Cout, k, dil=1, kind="fwd"):
    """(tile rows, tile columns, split factor) of the bf16 launch for this shape."""
    d = conv_desc(B, H, W, Cin, Cout, k, k, dil)
    out = (ctypes.c_int * 3)()
    _lib.call("emadapt_conv_bf16_plan", ctypes.byref(d), {"fwd": 0, "dgrad": 1, "wgrad": 2}[kind], out)
    return tuple(out)


def conv_executed_fraction(B, H, W, Cin, Cout, k, dil=1, kind="fwd", arithmetic="fp32"):
    """Executed / algorithmic FLOPs of the launch(es) for this shape: < 1 where the kernels skip a dilated filter's padding
    (host arithmetic only, no GPU needed)."""
    d = conv_desc(B, H, W, Cin, Cout, k, k, dil)
    f = ctypes.c_double(1.0)
    _lib.call("emadapt_conv_executed_fraction", ctypes.byref(d), {"fwd": 0, "dgrad": 1, "wgrad": 2}[kind],
              {"fp32": 0, "bf16": 1, "bf16x6": 2}[arithmetic], ctypes.byref(f))
    return f.value


def tune_set(key, value):
    """Force a launch-plan choice (include/emadapt.h "Launch-plan overrides"); 0 restores the automatic plan."""
    _lib.call("emadapt_tune_set", key.encode(), int(value))


def tune_get(key):
    v = ctypes.c_int(0)
    _lib.call("emadapt_tune_get", key.encode(), ctypes.byref(v))
    return v.value


class tuned:
    """Context manager: ``with ops.tuned(fp32_tile=2, plan_cus=4): ...`` -- restores the previous values on exit."""

    def __init__(self, **kv):
        self.kv = kv
        self.old = {}

    def __enter__(self):
        for k, v in self.kv.items():
            self.old[k] = tune_get(k)
            tune_set(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            tune_set(k, v)
        return False


def prof_enable(on=True):
    _lib.load().emadapt_prof_enable(1 if on else 0)


def prof_reset():
    _lib.load().emadapt_prof_reset()


def prof_collect():
    """{kernel class: dict(launches, ms, flops, flops_executed, bytes)} accumulated since the last reset.  flops = algorithmic
    (2*M*K*N); flops_executed = what the launches issued (less where a kernel skips a dilated filter's padding)."""
    n = len(_lib.K_NAMES)
    launches = (ctypes.c_int64 * n)()
    ms = (ctypes.c_double * n)()
    flops = (ctypes.c_double * n)()
    nbytes = (ctypes.c_double * n)()
    fexec = (ctypes.c_double * n)()
    _lib.call("emadapt_prof_collect2", launches, ms, flops, nbytes, fexec)
    return {name: dict(launches=int(launches[i]), ms=float(ms[i]), flops=float(flops[i]), flops_executed=float(fexec[i]),
                       bytes=float(nbytes[i]))
            for i, name in enumerate(_lib.K_NAMES)}
