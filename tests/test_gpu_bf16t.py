"""GPU parity of the LDS-DMA bf16 forward / data-gradient kernel (csrc/igemm_b16t.h, round 5: 8 MFMA waves on
v_mfma_f32_16x16x32_bf16 + 4 DMA waves; BASELINE configs[2] / [3] arithmetic: bf16 operands, fp32 accumulation) through the C ABI: the tiled filter image bit for bit against a NumPy restatement of its layout, then
forward (bias / ReLU / dropout mask) and data gradient (plain / gated) against the fp64 oracle evaluated on the same bf16-rounded
operands (2e-5: only the fp32 summation order differs), bf16 outputs = the f32 outputs rounded once, every result repeatable bit
for bit, in every launch layout -- plain tiles, stream-K with slab levels, whole rounds + stream-K tail -- forced onto small
awkward shapes by shrinking `plan_cus`; and at the benchmark shapes against the round-1 bf16 kernel."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available()
    import em_adapt_tensorflow_amd.ops as ops
    from oracle import deeplab_oracle as do
    return ops, do, torch


def _rel(got, want):
    want = np.asarray(want, dtype=np.float64)
    got = np.asarray(got, dtype=np.float64)
    return np.abs(got - want).max() / max(np.abs(want).max(), 1e-30)


def _tiles_ref(w, dgrad):
    """[tap][k / 32][rows padded to 256][4 slots x 8], slot s of row r = k chunk s ^ ((-(r >> 2)) & 3); bf16 bit patterns."""
    import torch
    KH, KW, Cin, Cout = w.shape
    wb = torch.tensor(w).bfloat16().view(torch.int16).numpy().reshape(KH * KW, Cin, Cout)
    m = wb.transpose(0, 1, 2) if dgrad else wb.transpose(0, 2, 1)        # [tap][row][k]
    taps, rows, kdim = m.shape
    rp = -(-rows // 256) * 256
    out = np.zeros((taps, kdim // 32, rp, 4, 8), np.int16)
    for r in range(rows):
        for s in range(4):
            c = s ^ ((-(r >> 2)) & 3)
            out[:, :, r, s, :] = m[:, r, :].reshape(taps, kdim // 32, 4, 8)[:, :, c, :]
    return out.reshape(-1)


@pytest.mark.parametrize("shape", [(3, 3, 32, 40), (1, 1, 64, 264), (4, 4, 96, 8), (3, 3, 64, 32)])
def test_tiled_filter_image_layout(env, shape):
    ops, do, torch = env
    rng = np.random.default_rng(sum(shape))
    w = rng.normal(0, 1, shape).astype(np.float32)
    wg = torch.tensor(w, device="cuda:0")
    got = ops.filter_tiles_bf16(wg).view(torch.int16).cpu().numpy()
    assert np.array_equal(got, _tiles_ref(w, False))
    if shape[3] % 32 == 0:
        got = ops.filter_tiles_bf16(wg, dgrad=True).view(torch.int16).cpu().numpy()
        assert np.array_equal(got, _tiles_ref(w, True))


# (B, H, W, Cin, Cout, k, dil), plan_cus: M / N tails, several taps and dilations, K of 1 .. 144 steps, Cout not a multiple of 32
SMALL = [
    ((1, 7, 9, 32, 8, 3, 1), 256),
    ((2, 13, 11, 64, 40, 3, 2), 256),
    ((2, 19, 17, 32, 264, 1, 1), 3),       # two N tiles; 3 "CUs": stream-K pieces across tiles
    ((3, 21, 20, 96, 520, 3, 1), 4),       # 6 x 3 tiles on 4 CUs
    ((1, 41, 41, 1024, 1032, 1, 1), 10),   # N > 1024, 3.5 rounds of tiles: whole rounds as plain tiles + a stream-K tail (four and more rounds: plain)
    ((1, 41, 41, 64, 1032, 1, 1), 2),      # .. the same on 2 "CUs": 17.5 rounds, plain tiles all the way
    ((2, 16, 16, 128, 256, 4, 4), 5),      # 4x4 rate 4 (fc6's filter), 16 taps
    ((1, 25, 25, 32, 48, 3, 12), 256),     # rate 12 reaches far outside the map
    ((2, 30, 18, 160, 288, 3, 1), 7),
    ((2, 21, 20, 128, 320, 3, 1), 4),      # both directions stream-K
    ((1, 33, 31, 192, 64, 3, 2), 256),     # 6 K steps per tap: pieces of 1 .. 5 steps exercise all three bodies of the fragment rotation
    ((2, 40, 33, 64, 128, 3, 1), 3),       # 128 output columns in a 256-wide tile (zero filter rows); 11 M tiles on 3 "CUs": stream-K
    ((3, 57, 31, 128, 96, 3, 1), 256),     # 96 columns, 21 M tiles with a tail, plain tiles
    ((1, 70, 50, 32, 64, 3, 1), 2),        # 64 columns; 14 tiles on 2 "CUs"; 9 K steps per tile
    ((1, 16, 16, 32, 256, 1, 1), 256),     # ONE K step per tile: prologue and closing barriers only
]


@pytest.mark.parametrize("shape,cus", SMALL)
def test_b16t_small_shapes_every_layout(env, shape, cus):
    ops, do, torch = env
    B, H, W, Cin, Cout, k, dil = shape
    rng = np.random.default_rng(hash(shape) % (1 << 31))
    x = torch.tensor(rng.normal(0, 1, (B, H, W, Cin)).astype(np.float32)).bfloat16()
    w = torch.tensor(rng.normal(0, 0.2, (k, k, Cin, Cout)).astype(np.float32))
    b = torch.tensor(rng.normal(0, 0.5, (Cout,)).astype(np.float32))
    mask = torch.tensor((rng.random((B, H, W, Cout)) < 0.5).astype(np.uint8))
    xr = x.double().requires_grad_(True)
    wr = w.bfloat16().double()
    y = do.conv_same(xr, wr, b.double(), dil)
    xg, wg, bg, mg = x.cuda(), w.cuda(), b.cuda(), mask.cuda()
    wt = ops.filter_tiles_bf16(wg)
    with ops.tuned(plan_cus=cus):
        plan = ops.conv_bf16t_plan(B, H, W, Cin, Cout, k, dil)
        got = ops.conv_fwd_bf16t(xg, wt, (k, k), Cout, bg, dil, relu=False, out_dtype=torch.float32)
        assert _rel(got.cpu(), y.detach()) < 2e-5, plan
        assert torch.equal(got, ops.conv_fwd_bf16t(xg, wt, (k, k), Cout, bg, dil, relu=False, out_dtype=torch.float32)), plan
        assert _rel(ops.conv_fwd_bf16t(xg, wt, (k, k), Cout, None, dil, out_dtype=torch.float32).cpu(), (y - b.double()).detach()) < 2e-5
        want = torch.relu(y.detach()) * 2.0 * mask.double()
        got = ops.conv_fwd_bf16t(xg, wt, (k, k), Cout, bg, dil, relu=True, keep_prob=0.5, drop_mask=mg, out_dtype=torch.float32)
        assert _rel(got.cpu(), want) < 2e-5, plan
        if Cout % 8 == 0:
            assert torch.equal(ops.conv_fwd_bf16t(xg, wt, (k, k), Cout, bg, dil, relu=True, keep_prob=0.5, drop_mask=mg), got.bfloat16())
        # same sums as the round-1 kernel up to the order of the K steps
        wf, _ = ops.weights_to_bf16(wg, want_dgrad=False)
        old = ops.conv_fwd_bf16(xg, wf, (k, k), bg, dil, relu=True, keep_prob=0.5, drop_mask=mg)
        assert _rel(got.cpu(), old.cpu()) < 2e-5
    if Cout % 32 == 0:                       # data gradient: the roles of Cin / Cout swap
        dy = torch.tensor(rng.normal(0, 1, (B, H, W, Cout)).astype(np.float32)).bfloat16()
        gate = torch.tensor(rng.normal(0, 1, (B, H, W, Cin)).astype(np.float32)).bfloat16()
        y.backward(dy.double())
        dyg, gg = dy.cuda(), gate.cuda()
        wtd = ops.filter_tiles_bf16(wg, dgrad=True)
        with ops.tuned(plan_cus=cus):
            plan = ops.conv_bf16t_plan(B, H, W, Cin, Cout, k, dil, "dgrad")
            got = ops.conv_dgrad_bf16t(dyg, wtd, (k, k), Cin, dil, out_dtype=torch.float32)
            assert _rel(got.cpu(), xr.grad) < 2e-5, plan
            want = torch.where(gate.double() > 0, xr.grad * 2.0, torch.zeros((), dtype=torch.float64))
            g32 = ops.conv_dgrad_bf16t(dyg, wtd, (k, k), Cin, dil, gate=gg.float(), gate_scale=2.0)
            assert g32.dtype == torch.float32 and _rel(g32.cpu(), want) < 2e-5, plan
            assert torch.equal(g32, ops.conv_dgrad_bf16t(dyg, wtd, (k, k), Cin, dil, gate=gg.float(), gate_scale=2.0))
            assert torch.equal(ops.conv_dgrad_bf16t(dyg, wtd, (k, k), Cin, dil, gate=gg, gate_scale=2.0), g32.bfloat16())


# (B, H, W, Cin, Cout, k, dil), plan_cus: launches of ONE partial round whose 224-row tiles still fit the round (round 6, igemm_b16t.h M224:
# 8 MFMA waves as 2 x 4, 7 x 4 accumulator tiles each) -- M tails in the last tile (fewer than 112 rows: the second wave row idles; fewer than
# 16), N tails, every tap geometry, K of 1 .. 3 steps and long, and the model's shapes at batch 16 (conv4_x / conv5_x: 242 workgroups)
M224 = [
    ((3, 41, 41, 64, 256, 3, 1), 24),        # 5043 rows: 20 -> 23 tiles on 24 "CUs"; last tile 115 rows
    ((1, 30, 30, 32, 264, 1, 1), 10),        # two N tiles (one 8 wide), 4 -> 5 row tiles (the last: 4 rows); ONE K step
    ((2, 28, 32, 96, 512, 3, 2), 18),        # 1792 rows = 8 x 224 exactly: 7 -> 8 row tiles x 2
    ((1, 41, 41, 64, 128, 4, 4), 8),         # fc6's filter geometry; 128 columns in a 256-wide tile
    ((1, 40, 42, 32, 256, 3, 12), 8),        # rate 12
    ((16, 41, 41, 512, 512, 3, 1), 256),     # conv4_2 / conv4_3
    ((16, 41, 41, 512, 512, 3, 2), 256),     # conv5_x
]


@pytest.mark.parametrize("shape,cus", M224)
def test_b16t_224_row_tiles(env, shape, cus):
    """The 224-row tile variant against the 256-row tiles of the same launch (tune bf16t_sk = 3): every output element is the same sum
    in the same order (the K walk does not depend on the tile's shape), so forward and data gradient must agree BIT FOR BIT, with every
    epilogue; and against the fp64 oracle on rounded operands at the small shapes."""
    ops, do, torch = env
    B, H, W, Cin, Cout, k, dil = shape
    g = torch.Generator(device="cuda").manual_seed(B * 131 + H + Cout)
    xg = torch.randn((B, H, W, Cin), device="cuda", generator=g).relu_().bfloat16()
    wg = torch.randn((k, k, Cin, Cout), device="cuda", generator=g) * (2.0 / (k * k * Cin)) ** 0.5
    bg = torch.randn((Cout,), device="cuda", generator=g) * 0.1
    mg = (torch.rand((B, H, W, Cout), device="cuda", generator=g) < 0.5).to(torch.uint8)
    wt = ops.filter_tiles_bf16(wg)
    M = B * H * W
    with ops.tuned(plan_cus=cus):
        assert ops.conv_bf16t_plan(B, H, W, Cin, Cout, k, dil) == (M, 0, 0, 224)
        new = [ops.conv_fwd_bf16t(xg, wt, (k, k), Cout, bg, dil, relu=False, out_dtype=torch.float32),
               ops.conv_fwd_bf16t(xg, wt, (k, k), Cout, bg, dil, relu=True, keep_prob=0.5, drop_mask=mg),
               ops.conv_fwd_bf16t(xg, wt, (k, k), Cout, bg, dil, relu=True, keep_prob=0.5, drop_seed=11)]
        assert torch.equal(new[0], ops.conv_fwd_bf16t(xg, wt, (k, k), Cout, bg, dil, relu=False, out_dtype=torch.float32))
        with ops.tuned(bf16t_sk=3):
            assert ops.conv_bf16t_plan(B, H, W, Cin, Cout, k, dil)[3] != 224
            old = [ops.conv_fwd_bf16t(xg, wt, (k, k), Cout, bg, dil, relu=False, out_dtype=torch.float32),
                   ops.conv_fwd_bf16t(xg, wt, (k, k), Cout, bg, dil, relu=True, keep_prob=0.5, drop_mask=mg),
                   ops.conv_fwd_bf16t(xg, wt, (k, k), Cout, bg, dil, relu=True, keep_prob=0.5, drop_seed=11)]
    for a_, b_ in zip(new, old):
        assert torch.equal(a_, b_)
    if B < 16:
        y = do.conv_same(xg.cpu().double(), wg.cpu().bfloat16().double(), bg.cpu().double(), dil)
        assert _rel(new[0].cpu(), y) < 2e-5
    if Cout % 32 == 0:
        dyg = torch.randn((B, H, W, Cout), device="cuda", generator=g).bfloat16()
        wtd = ops.filter_tiles_bf16(wg, dgrad=True)
        with ops.tuned(plan_cus=cus):
            p = ops.conv_bf16t_plan(B, H, W, Cin, Cout, k, dil, "dgrad")
            nd = [ops.conv_dgrad_bf16t(dyg, wtd, (k, k), Cin, dil, out_dtype=torch.float32),
                  ops.conv_dgrad_bf16t(dyg, wtd, (k, k), Cin, dil, gate=xg, gate_scale=2.0)]
            with ops.tuned(bf16t_sk=3):
                od = [ops.conv_dgrad_bf16t(dyg, wtd, (k, k), Cin, dil, out_dtype=torch.float32),
                      ops.conv_dgrad_bf16t(dyg, wtd, (k, k), Cin, dil, gate=xg, gate_scale=2.0)]
        for a_, b_ in zip(nd, od):
            assert torch.equal(a_, b_), p


@pytest.mark.parametrize("shape", [(3, 256, 256), (1, 512, 4096), (4, 256, 512), (3, 512, 320), (1, 64, 96), (3, 288, 256)],
                         ids=lambda s: "k%d_%dto%d" % s)
def test_both_tiled_images_in_one_pass(env, shape):
    """emadapt_filter_tiles_bf16_both: one kernel where Cin % 256 == 0 and Cout % 256 == 0 (the model's layers), the two single-image
    kernels elsewhere -- the same bytes as two calls either way."""
    ops, do, torch = env
    k, Cin, Cout = shape
    g = torch.Generator().manual_seed(k * 1000 + Cin + Cout)
    w = torch.randn((k, k, Cin, Cout), generator=g).to("cuda:0")
    f0, d0 = ops.filter_tiles_bf16(w), ops.filter_tiles_bf16(w, dgrad=True)
    f1 = torch.full_like(f0, float("nan"))
    d1 = torch.full_like(d0, float("nan"))
    ops.filter_tiles_bf16_both(w, f1, d1)
    assert torch.equal(f0.view(torch.int16), f1.view(torch.int16)) and torch.equal(d0.view(torch.int16), d1.view(torch.int16))


def test_b16t_layout_coverage_of_the_small_shapes(env):
    """The forced layouts above really are the three kinds the benchmark uses."""
    ops, do, torch = env
    kinds = set()
    for shape, cus in SMALL:
        with ops.tuned(plan_cus=cus):
            B, H, W, Cin, Cout, k, dil = shape
            rows, sk, lv, tail = ops.conv_bf16t_plan(B, H, W, Cin, Cout, k, dil)
            kinds.add("sk%d" % min(lv, 2) if sk else ("plain" if rows == B * H * W else "rounds+tail%d" % min(tail, 1)))
    assert {"plain", "sk2"} <= kinds and any(k.startswith("rounds+tail") for k in kinds), kinds


def test_unsupported_shapes_are_refused(env):
    ops, do, torch = env
    x = torch.zeros((1, 8, 8, 24), dtype=torch.bfloat16, device="cuda:0")
    d = ops.conv_desc(1, 8, 8, 24, 32, 3, 3, 1)
    assert not ops.conv_bf16t_supported(d, "fwd") and ops.conv_bf16t_supported(d, "dgrad")
    with pytest.raises(ValueError):
        ops.filter_tiles_bf16(torch.zeros((3, 3, 24, 32), device="cuda:0"))
    with pytest.raises(TypeError):
        ops.conv_fwd_bf16t(x.float(), torch.zeros(8, dtype=torch.bfloat16, device="cuda:0"), (3, 3), 32)
