"""CPU oracle for the DeepLab graph path of EM-Adapt.  TEST INFRASTRUCTURE ONLY.

A restatement, on PyTorch-CPU fp32/fp64 ops, of what the reference builds out of TensorFlow 1.x in
deeplab.py:35-107 (network), :109-124 (E-step glue), :180-186 (loss), :188-208 + network.py:91-102
(gradient accumulation + Momentum) and :265-270 (one micro-step).

PARITY UNPINNED for this part: TensorFlow 1.x is an un-vendored, un-pinned third-party dependency that
is absent here (no `tensorflow` module, no network), and the reference holds no tests or golden
vectors for it.  The TF semantics encoded below are the documented ones:
  * conv2d / atrous_conv2d: cross-correlation, HWIO filters, stride 1, SAME = zero padding of
    (dilated_k-1)//2 before and the rest after (3x3 rate r -> r/r; 4x4 rate 4 -> 6/6, taps -6,-2,+2,+6);
  * max_pool 3x3 SAME: out = ceil(in/stride), padding never wins (treated as -inf), gradient to the
    first maximum of the window;
  * dropout: x / keep_prob * mask (TF <= 1.12; from 1.13 on x * (1 / keep_prob) * mask, which is what the HIP kernels compute --
    the same number for keep_prob = 2^-n, i.e. the reference's 0.5, deeplab.py:266);  sparse_softmax_cross_entropy + reduce_mean over all B*H*W pixels;
  * l2_loss = sum(w^2)/2 over filter weights only;  MomentumOptimizer non-Nesterov: v = mu*v + g, w -= lr*v;
  * the LR multipliers of deeplab.py:194-200 are dead code (the loop rebinds a local) -> none applied;
  * the E-step sees a copy of the logits; the loss uses the untouched fc8 output (deeplab.py:182).
The E-step itself IS pinned (oracle/estep_oracle.py vs goldens from the reference's estep.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

from . import estep_oracle as _eo


def layer_specs(category_num=21, preset="ref", width=64, fc_width=None, input_channel=3):
    """[(name, kind, dict)] in build order -- deeplab.py:40-45 with shapes from :131-149.

    preset "ref": the network as written (fc6 4x4 rate 4); "largefov": fc6 3x3 rate 12 (DeepLab-LargeFOV).
    `width` scales every channel count (64 in the reference) so that tests can run a narrow net.
    """
    if fc_width is None:
        fc_width = (4096 if preset == "ref" else 1024) * width // 64
    specs = []
    cin = input_channel
    stride = 1
    blocks = [(2, 1), (2, 1), (3, 1), (3, 1), (3, 2)]  # (convs, rate); conv5_x is atrous rate 2 (deeplab.py:65)
    for bi, (n, rate) in enumerate(blocks, start=1):
        cout = min(width * stride, width * 8)
        for ci in range(1, n + 1):
            specs.append(("conv%d_%d" % (bi, ci), "conv", dict(k=3, cin=cin, cout=cout, rate=rate, relu=True)))
            cin = cout
        pstride = 2 if bi <= 3 else 1  # deeplab.py:74-83
        specs.append(("pool%d" % bi, "pool", dict(stride=pstride)))
        if pstride == 2:
            stride *= 2
    if preset == "ref":
        specs.append(("fc6", "conv", dict(k=4, cin=cin, cout=fc_width, rate=4, relu=True, drop=True)))
    elif preset == "largefov":
        specs.append(("fc6", "conv", dict(k=3, cin=cin, cout=fc_width, rate=12, relu=True, drop=True)))
    else:
        raise ValueError(preset)
    specs.append(("fc7", "conv", dict(k=1, cin=fc_width, cout=fc_width, rate=1, relu=True, drop=True)))
    specs.append(("fc8", "conv", dict(k=1, cin=fc_width, cout=category_num, rate=1, relu=False)))
    return specs


def init_params(specs, seed=2, std=0.01, he=False):
    """deeplab.py:150-154: W ~ N(0, std^2), b = 0.  `he=True` uses sqrt(2/fan_in) so that logits have
    spread (parity tests); returns {layer: (w HWIO float32, b float32)} as NumPy arrays."""
    rng = np.random.default_rng(seed)
    params = {}
    for name, kind, p in specs:
        if kind != "conv":
            continue
        shape = (p["k"], p["k"], p["cin"], p["cout"])
        s = math.sqrt(2.0 / (p["k"] * p["k"] * p["cin"])) if he else std
        w = rng.normal(0.0, s, shape).astype(np.float32)
        b = (rng.normal(0.0, 0.05, (p["cout"],)).astype(np.float32) if he else np.zeros((p["cout"],), np.float32))
        params[name] = (w, b)
    return params


def conv_same(x, w, b, rate):
    """x [B,H,W,Cin], w HWIO, SAME stride 1 (dilated) + bias -> [B,H,W,Cout]."""
    kh, kw = w.shape[0], w.shape[1]
    eh, ew = (kh - 1) * rate, (kw - 1) * rate
    pt, pl = eh // 2, ew // 2
    xn = x.permute(0, 3, 1, 2)
    xn = F.pad(xn, (pl, ew - pl, pt, eh - pt))
    y = F.conv2d(xn, w.permute(3, 2, 0, 1), b, stride=1, padding=0, dilation=rate)
    return y.permute(0, 2, 3, 1)


def _bf16_round(t):
    """Round to bf16 (round-to-nearest-even, as v_cvt_pk_bf16_f32) and return in t's dtype."""
    return t.to(torch.float32).bfloat16().to(t.dtype)


class _ConvBf16(torch.autograd.Function):
    """Arithmetic model of the bf16-compute kernels (csrc/conv_bf16.hip): every inner product -- forward, data gradient
    and filter gradient -- takes both operands rounded to bf16 and accumulates exactly (here: in the tensor dtype)."""

    @staticmethod
    def forward(ctx, x, w, rate):
        ctx.save_for_backward(x, w)
        ctx.rate = rate
        return conv_same(_bf16_round(x), _bf16_round(w), None, rate)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dyr = _bf16_round(dy)
        with torch.enable_grad():
            xr = _bf16_round(x).detach().requires_grad_(True)
            wr = _bf16_round(w).detach().requires_grad_(True)
            y = conv_same(xr, wr, None, ctx.rate)
            dx, dw = torch.autograd.grad(y, (xr, wr), dyr)
        return dx, dw, None


class _RoundFwd(torch.autograd.Function):
    """An activation stored as bf16: rounded on the way forward, gradient passes unchanged."""

    @staticmethod
    def forward(ctx, x):
        return _bf16_round(x)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundBwd(torch.autograd.Function):
    """An activation GRADIENT stored as bf16: identity forward, the gradient is rounded on the way back."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return _bf16_round(g)


def conv_same_bf16(x, w, b, rate):
    y = _ConvBf16.apply(x, w, rate)
    return y if b is None else y + b


def maxpool_same(x, stride):
    B, H, W, C = x.shape
    oh, ow = -(-H // stride), -(-W // stride)
    th = max((oh - 1) * stride + 3 - H, 0)
    tw = max((ow - 1) * stride + 3 - W, 0)
    xn = x.permute(0, 3, 1, 2)
    xn = F.pad(xn, (tw // 2, tw - tw // 2, th // 2, th - th // 2), value=float("-inf"))
    y = F.max_pool2d(xn, kernel_size=3, stride=stride, padding=0)
    return y.permute(0, 2, 3, 1)


def output_size(h, specs):
    for _, kind, p in specs:
        if kind == "pool":
            h = -(-h // p["stride"])
    return h


class DeepLabOracle:
    """Forward / loss / backward / update of one micro-step, on the CPU."""

    def __init__(self, category_num=21, preset="ref", width=64, fc_width=None, input_channel=3,
                 dtype=torch.float32, compute_dtype="fp32", act_dtype=None):
        self.C = category_num
        self.conv = conv_same_bf16 if compute_dtype == "bf16" else conv_same
        # act_dtype "bf16" (default with bf16 compute): activations are rounded when stored (after bias/ReLU/dropout), and
        # activation gradients when stored (after the ReLU / dropout gate); the logits stay unrounded, their gradient not
        self.store_bf16 = (act_dtype or compute_dtype) == "bf16"
        self.specs = layer_specs(category_num, preset, width, fc_width, input_channel)
        self.dtype = dtype
        self.estep_args = dict(bg_p=0.4, fg_p=0.2, num_iter=5, suppress_others=True, margin_others=1e-5)  # deeplab.py:181

    def tensors(self, params, requires_grad=True):
        out = {}
        for k, (w, b) in params.items():
            out[k] = (torch.tensor(w, dtype=self.dtype, requires_grad=requires_grad),
                      torch.tensor(b, dtype=self.dtype, requires_grad=requires_grad))
        return out

    def forward(self, tparams, images, keep_prob=1.0, masks=None, keep_all=False):
        """images [B,H,W,3]; masks {"fc6": uint8/bool [B,h,w,C], "fc7": ...} injected keep masks."""
        x = torch.as_tensor(images, dtype=self.dtype)
        acts = {"input": x}
        last = self.specs[-1][0]
        for name, kind, p in self.specs:
            if kind == "conv":
                w, b = tparams[name]
                x = self.conv(x, w, b, p["rate"])
                if self.store_bf16:
                    x = _RoundBwd.apply(x)
                if p.get("relu"):
                    x = torch.relu(x)
                if p.get("drop") and keep_prob < 1.0:
                    m = torch.as_tensor(np.asarray(masks[name]), dtype=self.dtype)
                    x = x / keep_prob * m  # tf.nn.dropout: div then multiply by the binary mask
                if self.store_bf16 and name != last:
                    x = _RoundFwd.apply(x)
            else:
                x = maxpool_same(x, p["stride"])
            if keep_all:
                acts[name] = x
        acts["output"] = x
        return acts

    def estep_labels(self, logits_np, label_full, perms):
        """deeplab.py:109-124: nearest-resize the pixel labels to the logit grid, E-step, argmax."""
        B, S1, S2, C = logits_np.shape
        shrink = _eo.resize_nearest_oracle(label_full, (S1, S2))
        _, labels = _eo.estep_oracle(logits_np.astype(np.float32), shrink, perms, **self.estep_args)
        return labels, shrink

    def micro_step(self, params, images, label_full, perms, weight_decay=1e-5, keep_prob=1.0, masks=None,
                   labels_override=None):
        """One fwd + E-step + bwd.  Returns dict(loss_norm, loss_l2, logits, labels, grads{layer:(dw,db)}).
        grads are d(norm + wd*l2)/d(param), i.e. what compute_gradients yields (deeplab.py:192)."""
        tp = self.tensors(params)
        acts = self.forward(tp, images, keep_prob, masks)
        logits = acts["output"]
        lnp = logits.detach().to(torch.float32).numpy()
        import time as _time
        t_e = _time.perf_counter()
        if labels_override is None:
            labels, _ = self.estep_labels(lnp, label_full, perms)
        else:
            labels = labels_override
        t_e = _time.perf_counter() - t_e
        lt = torch.as_tensor(labels.reshape(-1), dtype=torch.int64)
        norm = F.cross_entropy(logits.reshape(-1, self.C), lt, reduction="mean")
        l2 = sum((w * w).sum() / 2 for (w, _) in tp.values())
        total = norm + weight_decay * l2
        total.backward()
        grads = {k: (w.grad.detach().numpy().copy(), b.grad.detach().numpy().copy()) for k, (w, b) in tp.items()}
        return dict(loss_norm=float(norm.detach()), loss_l2=float(l2.detach()), logits=lnp, labels=labels, grads=grads,
                    estep_seconds=t_e)


def momentum_update(params, velocity, accum, lr, momentum):
    """apply_gradients of tf.train.MomentumOptimizer on the accumulators, then the clean op
    (deeplab.py:203-205); in place on dicts of NumPy arrays {layer: [w, b]}."""
    for k in params:
        for i in range(2):
            velocity[k][i] = (momentum * velocity[k][i] + accum[k][i]).astype(np.float32)
            params[k][i] = (params[k][i] - np.float32(lr) * velocity[k][i]).astype(np.float32)
            accum[k][i][...] = 0


def synthetic_batch(B, size, category_num, seed=0, max_objects=3):
    """SURVEY section 8(d) inputs: images ~ N(0, 50^2); pixel labels = background 0 + up to 3 rectangles
    (side >= size/5) of random foreground classes + a 4-pixel void (255) border."""
    rng = np.random.default_rng(seed)
    images = rng.normal(0.0, 50.0, (B, size, size, 3)).astype(np.float32)
    labels = np.zeros((B, size, size, 1), np.float32)
    lo = max(size // 5, 2)
    for b in range(B):
        for _ in range(int(rng.integers(1, max_objects + 1))):
            c = int(rng.integers(1, category_num))
            h = int(rng.integers(lo, max(size // 2, lo + 1)))
            w = int(rng.integers(lo, max(size // 2, lo + 1)))
            y = int(rng.integers(0, size - h + 1))
            x = int(rng.integers(0, size - w + 1))
            labels[b, y:y + h, x:x + w, 0] = c
        brd = min(4, size // 16)
        if brd:
            labels[b, :brd] = 255
            labels[b, -brd:] = 255
            labels[b, :, :brd] = 255
            labels[b, :, -brd:] = 255
    return images, labels


def resize_bilinear_tf1(x, out_hw):
    """tf.image.resize_bilinear(x, out_hw), align_corners=False, TF 1.x legacy kernel [TF-doc]: source coordinate =
    dst * (in/out) (no half-pixel centre), lerp between floor and min(floor+1, in-1); fp32 arithmetic in this order:
    top = tl + (tr - tl)*xl; bottom = bl + (br - bl)*xl; out = top + (bottom - top)*yl.  x: [B,H,W,C] float32."""
    x = np.asarray(x, dtype=np.float32)
    B, H, W, C = x.shape
    oh, ow = out_hw
    f32 = np.float32
    ys = (np.arange(oh, dtype=np.float32) * (f32(H) / f32(oh))).astype(np.float32)
    xs = (np.arange(ow, dtype=np.float32) * (f32(W) / f32(ow))).astype(np.float32)
    y0 = np.floor(ys).astype(np.int64); x0 = np.floor(xs).astype(np.int64)
    y1 = np.minimum(y0 + 1, H - 1); x1 = np.minimum(x0 + 1, W - 1)
    yl = (ys - y0.astype(np.float32)).astype(np.float32)[None, :, None, None]
    xl = (xs - x0.astype(np.float32)).astype(np.float32)[None, None, :, None]
    tl = x[:, y0][:, :, x0]; tr = x[:, y0][:, :, x1]; bl = x[:, y1][:, :, x0]; br = x[:, y1][:, :, x1]
    top = (tl + ((tr - tl) * xl).astype(np.float32)).astype(np.float32)
    bot = (bl + ((br - bl) * xl).astype(np.float32)).astype(np.float32)
    return (top + ((bot - top) * yl).astype(np.float32)).astype(np.float32)


def image_preprocess_oracle(img_u8, label_u8, out_hw, scale=None, flip=False):
    """dataset.py:147-185 restated in NumPy (TF1 semantics [TF-doc]): optional random scale (bilinear image / nearest
    label to int(h*scale), int(w*scale)), resize to out_hw, RGB->BGR, float32, minus mean, optional left-right flip.
    The uint8 `label -= 255 ... += 255` wrap-around of the reference is the identity under nearest resizing."""
    img = np.asarray(img_u8, dtype=np.float32)[None]
    lab = np.asarray(label_u8, dtype=np.float32).reshape(1, img.shape[1], img.shape[2])
    if scale is not None:
        hn = int(np.float32(img.shape[1]) * np.float32(scale))
        wn = int(np.float32(img.shape[2]) * np.float32(scale))
        img = resize_bilinear_tf1(img, (hn, wn))
        lab = _eo.resize_nearest_oracle(lab, (hn, wn))
    img = resize_bilinear_tf1(img, out_hw)
    lab = _eo.resize_nearest_oracle(lab, out_hw)
    mean = np.array([104.00698793, 116.66876762, 122.67891434], np.float32)
    out = img[0][:, :, ::-1].astype(np.float32) - mean
    lab = lab[0].astype(np.float32)
    if flip:
        out = out[:, ::-1]
        lab = lab[:, ::-1]
    return np.ascontiguousarray(out, np.float32), np.ascontiguousarray(lab)[..., None]
