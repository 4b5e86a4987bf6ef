"""GPU parity of the bf16 conv kernels AT THE BENCHMARK SHAPES of BASELINE configs[2] / [3] (16 images per GPU, bf16 operands,
fp32 accumulation): fc6, fc7, conv4_2, conv3_2 -- the 256x256x64 one-workgroup-per-CU launches (1696 workgroups for fc6 / fc7) --
and conv1_2 (12,881 workgroups of 128x128) run forward with the model's epilogues (bias + ReLU [+ dropout mask]:
deeplab.py:58-59,92-100), data gradient with the ReLU / dropout gate and filter gradient, against the fp64 oracle evaluated on
the SAME bf16-rounded operands (products of bf16 numbers are exact in fp32, so only the fp32 summation order differs: 2e-5 of the
tensor's max magnitude), in both storage types (bf16-stored tensors must give bit-identical sums to f32-stored tensors holding the
same values; a bf16 output is the f32 output rounded once), repeatable bit for bit, with the launch plan asserted
(emadapt_conv_bf16_plan).  The small-shape file (test_gpu_bf16.py) never reaches these grids."""
import os
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available()
    import em_adapt_tensorflow_amd.ops as ops
    from oracle import deeplab_oracle as do
    torch.set_num_threads(max(1, min(64, len(os.sched_getaffinity(0)))))
    return ops, do, torch


def _rel(got, want):
    want = np.asarray(want, dtype=np.float64)
    got = np.asarray(got, dtype=np.float64)
    return np.abs(got - want).max() / max(np.abs(want).max(), 1e-30)


@pytest.fixture
def implicit_gemm_kernel():
    """tune bf16_tile = 128: conv1_2 on the implicit-GEMM kernel (what it ran on before the direct kernel, and still runs on with f32 tensors)."""
    import em_adapt_tensorflow_amd.ops as ops
    ops.tune_set("bf16_tile", 128)
    yield
    ops.tune_set("bf16_tile", 0)


# name, (B, H, W, Cin, Cout, k, dil), tile of forward / data gradient, tile of the filter gradient, dropout
FULL = [
    ("fc6", (16, 41, 41, 512, 4096, 4, 4), 256, (256, 256), True),      # deeplab.py:92,145
    ("fc7", (16, 41, 41, 4096, 4096, 1, 1), 256, (256, 256), True),     # deeplab.py:95,147
    ("conv4_2", (16, 41, 41, 512, 512, 3, 1), 256, (256, 256), False),
    ("conv3_2", (16, 81, 81, 256, 256, 3, 1), 256, (256, 256), False),
    ("conv1_2", (16, 321, 321, 64, 64, 3, 1), 128, (128, 64), False),   # 12,881 workgroups; 1.65 M pixels (forced: tune bf16_tile = 128)
    ("conv1_2 direct", (16, 321, 321, 64, 64, 3, 1), 4, (4, 16), False),    # the automatic plan: the direct kernel (csrc/conv_d64.h), 7,056 tiles of 16 x 16
    ("conv2_2", (16, 161, 161, 128, 128, 3, 1), 128, (128, 128), False),    # the automatic plan of the 128-column layers: 128 x 128 tiles, 3,241 workgroups
    # the other layers still on the register-staged kernels (round-5 verdict: parity rows at batch 16); (forward tile, data-gradient tile)
    ("conv2_1", (16, 161, 161, 64, 128, 3, 1), (128, 128), (128, 128), False),
    ("conv3_1", (16, 81, 81, 128, 256, 3, 1), (256, 128), (256, 256), False),
    ("conv4_1", (16, 41, 41, 256, 512, 3, 1), (256, 128), (256, 256), False),
]


@pytest.mark.parametrize("name,shape,tile,wtile,drop", FULL, ids=[f[0] for f in FULL])
def test_bf16_benchmark_shape_against_fp64_oracle_on_rounded_operands(env, request, name, shape, tile, wtile, drop):
    ops, do, torch = env
    B, H, W, Cin, Cout, k, dil = shape
    if name == "conv1_2":
        request.getfixturevalue("implicit_gemm_kernel")
    pf, pd, pw = (ops.conv_bf16_plan(B, H, W, Cin, Cout, k, dil, kind) for kind in ("fwd", "dgrad", "wgrad"))
    tf_, td_ = tile if isinstance(tile, tuple) else (tile, tile)
    assert pf[0] == tf_ and pd[0] == td_ and (pw[:2] == wtile or name == "conv1_2"), (pf, pd, pw)   # (a forced 128 keeps the filter gradient's 128-wide tile)
    if name == "fc6":
        assert pd[2] > 1, pd          # 212 tiles of a 65536-long reduction: split-K slabs

    g = torch.Generator().manual_seed(zlib.crc32(name.encode()))
    fan_in = k * k * Cin
    # operands ARE bf16 values (what the model stores): post-ReLU input, He filters, unit gradients
    x = torch.randn((B, H, W, Cin), generator=g).clamp_(min=0).bfloat16()
    w = (torch.randn((k, k, Cin, Cout), generator=g) * float(np.sqrt(2.0 / fan_in))).float()
    b = (torch.randn((Cout,), generator=g) * 0.1).float()
    dy = torch.randn((B, H, W, Cout), generator=g).bfloat16()
    mask = (torch.rand((B, H, W, Cout), generator=g) < 0.5).to(torch.uint8) if drop else None

    # fp64 oracle on the rounded operands
    xr = x.double().requires_grad_(True)
    wr = w.bfloat16().double().requires_grad_(True)
    y_lin = do.conv_same(xr, wr, b.double(), dil)
    y_lin.backward(dy.double())
    want_y = torch.relu(y_lin.detach())
    scale = 2.0 if drop else 1.0
    if drop:
        want_y = want_y * scale * mask.double()
    want_dx = torch.where(x.double() > 0, xr.grad * scale, torch.zeros((), dtype=torch.float64))
    want_dw = wr.grad
    del y_lin

    dev = "cuda:0"
    xg, wg, bg, dyg = x.to(dev), w.to(dev), b.to(dev), dy.to(dev)
    mg = mask.to(dev) if drop else None
    wf, wd = ops.weights_to_bf16(wg)
    kw = dict(relu=True, keep_prob=0.5 if drop else 1.0, drop_mask=mg)

    if tile == 4:
        # the direct kernel takes bf16-stored x, y and gate (what the model keeps); every other storage combination runs the implicit-GEMM
        # kernel -- the same function, another order of the fp32 sums: one bf16 ulp apart at most
        def close(got, want, tol):
            err = (got.double().cpu() - want.double().cpu()).abs()
            return bool((err <= want.double().cpu().abs() * tol + 3e-5).all())
        yb = ops.conv_fwd_bf16(xg, wf, (k, k), bg, dil, out_dtype=torch.bfloat16, **kw)
        assert close(yb, want_y, 2.0 ** -8)
        assert torch.equal(yb, ops.conv_fwd_bf16(xg, wf, (k, k), bg, dil, out_dtype=torch.bfloat16, **kw)), "forward must repeat bitwise"
        y32 = ops.conv_fwd_bf16(xg, wf, (k, k), bg, dil, **kw)   # f32 output: the implicit-GEMM kernel
        assert _rel(y32.cpu(), want_y) < 2e-5 and close(yb, y32.bfloat16(), 2.0 ** -7) and (yb == y32.bfloat16()).float().mean().item() > 0.99
        del yb, y32
        dxb = ops.conv_dgrad_bf16(dyg, wd, (k, k), dil, gate=xg, gate_scale=scale)
        assert dxb.dtype == torch.bfloat16 and close(dxb, want_dx, 2.0 ** -8)
        assert torch.equal(dxb, ops.conv_dgrad_bf16(dyg, wd, (k, k), dil, gate=xg, gate_scale=scale))
        dx32 = ops.conv_dgrad_bf16(dyg, wd, (k, k), dil, gate=xg.float(), gate_scale=scale)
        assert _rel(dx32.cpu(), want_dx) < 2e-5 and close(dxb, dx32.bfloat16(), 2.0 ** -7) and (dxb == dx32.bfloat16()).float().mean().item() > 0.99
        del dxb, dx32
        # the filter gradient on its direct kernel (csrc/wgrad_d64.h: 256 slabs of 9 x 64 x 64 + the bias row)
        dw1 = ops.conv_wgrad_bf16(xg, dyg, k, k, dil)
        assert _rel(dw1.cpu(), want_dw) < 2e-5
        assert torch.equal(dw1, ops.conv_wgrad_bf16(xg, dyg, k, k, dil)), "filter gradient must repeat bitwise"
        db = torch.empty((Cout,), device=dev)
        assert torch.equal(ops.conv_wgrad_bf16(xg, dyg, k, k, dil, bias_out=db), dw1)
        assert _rel(db.cpu().double(), dy.double().sum(dim=(0, 1, 2))) < 1e-5
        assert _rel(ops.conv_wgrad_bf16(xg.float(), dyg.float(), k, k, dil).cpu(), dw1.cpu()) < 2e-5   # (f32-stored operands: the split-M kernel)
        return
    # forward: bf16-stored input -> f32 output
    y1 = ops.conv_fwd_bf16(xg, wf, (k, k), bg, dil, **kw)
    assert _rel(y1.cpu(), want_y) < 2e-5
    assert torch.equal(y1, ops.conv_fwd_bf16(xg, wf, (k, k), bg, dil, **kw)), "forward must repeat bitwise"
    assert torch.equal(y1, ops.conv_fwd_bf16(xg.float(), wf, (k, k), bg, dil, **kw)), "f32-stored input: the same sums"
    assert torch.equal(ops.conv_fwd_bf16(xg, wf, (k, k), bg, dil, out_dtype=torch.bfloat16, **kw), y1.bfloat16())
    del y1

    # data gradient with the ReLU (x dropout scale) gate of the layer below
    dx1 = ops.conv_dgrad_bf16(dyg, wd, (k, k), dil, gate=xg.float(), gate_scale=scale)
    assert _rel(dx1.cpu(), want_dx) < 2e-5
    assert torch.equal(dx1, ops.conv_dgrad_bf16(dyg, wd, (k, k), dil, gate=xg.float(), gate_scale=scale))
    assert torch.equal(dx1, ops.conv_dgrad_bf16(dyg.float(), wd, (k, k), dil, gate=xg.float(), gate_scale=scale))
    assert torch.equal(ops.conv_dgrad_bf16(dyg, wd, (k, k), dil, gate=xg, gate_scale=scale), dx1.bfloat16())   # bf16 gate -> bf16 dx
    del dx1

    # the LDS-DMA kernel (csrc/igemm_b16t.h: 8 MFMA waves on v_mfma_f32_16x16x32_bf16 + 4 DMA waves, round 5) at the same shapes:
    # whole rounds + stream-K tail (fc6 / fc7 forward) resp. stream-K over everything, f32 and bf16 outputs
    d = ops.conv_desc(B, H, W, Cin, Cout, k, k, dil)
    if ops.conv_bf16t_supported(d, "fwd") and ops.conv_bf16t_supported(d, "dgrad"):
        wt, wtd = ops.filter_tiles_bf16(wg), ops.filter_tiles_bf16(wg, dgrad=True)
        yt = ops.conv_fwd_bf16t(xg, wt, (k, k), Cout, bg, dil, out_dtype=torch.float32, **kw)
        assert _rel(yt.cpu(), want_y) < 2e-5
        assert torch.equal(yt, ops.conv_fwd_bf16t(xg, wt, (k, k), Cout, bg, dil, out_dtype=torch.float32, **kw))
        assert torch.equal(ops.conv_fwd_bf16t(xg, wt, (k, k), Cout, bg, dil, **kw), yt.bfloat16())
        dxt = ops.conv_dgrad_bf16t(dyg, wtd, (k, k), Cin, dil, gate=xg.float(), gate_scale=scale)
        assert _rel(dxt.cpu(), want_dx) < 2e-5
        assert torch.equal(ops.conv_dgrad_bf16t(dyg, wtd, (k, k), Cin, dil, gate=xg, gate_scale=scale), dxt.bfloat16())
        del yt, dxt

    # filter gradient
    dw1 = ops.conv_wgrad_bf16(xg, dyg, k, k, dil)
    assert _rel(dw1.cpu(), want_dw) < 2e-5
    assert torch.equal(dw1, ops.conv_wgrad_bf16(xg, dyg, k, k, dil)), "filter gradient must repeat bitwise"
    assert torch.equal(dw1, ops.conv_wgrad_bf16(xg.float(), dyg.float(), k, k, dil))
