"""Data-parallel step on the GPU: two ranks (sharing the one visible GPU, gloo transport for the rehearsal --
the production backend is nccl/RCCL, one rank per GPU) run micro_step on their half of a batch, exchange the
accumulators through GradSync and apply Momentum; the weights must match a single-process run on the whole batch
(the reference's ``accum += g/accum_num`` semantics with micro-batches mapped to ranks, deeplab.py:202)."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = dict(size=65, C=5, width=8, B=2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _inputs():
    from oracle import deeplab_oracle as do
    from oracle import estep_oracle as eo
    images, labels = do.synthetic_batch(2 * CFG["B"], CFG["size"], CFG["C"], seed=31)
    oracle = do.DeepLabOracle(category_num=CFG["C"], width=CFG["width"])
    params = do.init_params(oracle.specs, seed=6, he=True)
    return (images / 50).astype(np.float32), labels, params, eo.draw_perms(5, CFG["C"], seed=2)


def _build(torch, params, B, dtype="fp32", exchange=None):
    from em_adapt_tensorflow_amd.deeplab import ADAPT
    cfg = {"input_size": (CFG["size"], CFG["size"]), "category_num": CFG["C"], "width": CFG["width"],
           "batch_size": B, "accum_num": 1, "device": "cuda:0", "bucket_mb": 1, "compute_dtype": dtype}
    if dtype == "bf16x6":
        # at this size and width the automatic plan leaves every layer to the native kernels (advisor finding, round 3): force
        # the plane kernels so that the parametrisation tests what its name says
        cfg["x6_planes"] = "always"
    if exchange:
        cfg.update(exchange)
    a = ADAPT(cfg)
    a.build()
    if dtype == "bf16x6":
        assert any(v["fwd"] or v["wgrad"] for v in a._alloc(B)["xp"].values()), "no layer runs on the bf16x6 plane kernels"
    for layer, (w, b) in params.items():
        a.weights[layer][0].copy_(torch.tensor(w).cuda())
        a.weights[layer][1].copy_(torch.tensor(b).cuda())
    a.pre_train(base_lr=1e-2, weight_decay=1e-3, momentum=0.9, batch_size=B)
    return a


def _worker(rank, world, port, out_dir, dtype, exchange=None):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    images, labels, params, perms = _inputs()
    B = CFG["B"]
    a = _build(torch, params, B, dtype, exchange)
    assert a._gradsync is not None and a._gradsync.world_size == world and len(a._gradsync.buckets) > 1
    if exchange:
        assert a._gradsync.algo == "direct" and a._gradsync.payload == exchange.get("grad_dtype", "fp32")
        if "bucket_mb" in exchange and exchange["bucket_mb"] is None:
            fc_end = max(op.seg["end"] for op in a._plan if op.kind == "conv" and op.name.startswith("fc"))
            assert a._gradsync.buckets == [(0, fc_end // world * world), (fc_end // world * world, a._gradsync.total)]
    a._gradsync.timing = True   # off by default (a training run must not pile up events); stats() below reads them
    sl = slice(rank * B, (rank + 1) * B)
    for _ in range(2):  # two updates: the second exercises momentum and the re-armed GradSync
        a.micro_step(images[sl], labels[sl], keep_prob=1.0, perms=perms)
        a.apply_update()
    torch.cuda.synchronize()
    st = a._gradsync.stats()
    assert st["updates"] == 2 and st["exposed_ms"] is not None and st["exposed_ms"] >= 0 and st["bytes_per_update"] > 0
    np.save(os.path.join(out_dir, "params%d.npy" % rank), a._flat["params"].cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("dtype,bar", [("fp32", 1e-5), ("bf16", 2e-4), ("bf16x6", 2e-5)])   # bf16 = BASELINE configs[2] / [3]'s arithmetic; bf16x6 = the fp32-grade fast path
def test_two_ranks_match_single_process(tmp_path, dtype, bar):
    import torch
    import torch.multiprocessing as mp
    assert torch.cuda.is_available()
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), dtype), nprocs=2, join=True)
    p0 = np.load(tmp_path / "params0.npy")
    p1 = np.load(tmp_path / "params1.npy")
    assert np.array_equal(p0, p1), "ranks diverged"
    # single process, whole batch
    images, labels, params, perms = _inputs()
    a = _build(torch, params, 2 * CFG["B"], dtype)
    assert a._gradsync is None
    for _ in range(2):
        a.micro_step(images, labels, keep_prob=1.0, perms=perms)
        a.apply_update()
    torch.cuda.synchronize()
    want = a._flat["params"].cpu().numpy()
    err = np.abs(p0 - want).max() / np.abs(want).max()
    assert err < bar, err


@pytest.mark.parametrize("dtype,exchange,bar", [("fp32", {"allreduce_algo": "direct"}, 1e-5),
                                               ("bf16", {"grad_dtype": "bf16"}, 2e-3),
                                               # (no bucket size given: the two-bucket plan, fc8 + fc7 + fc6 | the convolutions)
                                               ("bf16", {"grad_dtype": "bf16", "bucket_mb": None}, 2e-3)])
def test_direct_exchange_and_bf16_gradient_payload(tmp_path, dtype, exchange, bar):
    """The all-links exchange (reduce-scatter + all-gather as two all-to-all steps on a side stream, fp32 accumulation by the
    shard's owner) in fp32, and with the bf16 gradient payload beside bf16 compute (configs[2] / [3]): ranks bit-identical, and
    the weights after two updates equal the single-process whole-batch run -- to fp32 rounding / to the bf16 rounding of the
    exchanged gradient (two roundings of 2^-9 on gradients that move a weight by lr x g)."""
    import torch
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), dtype, exchange), nprocs=2, join=True)
    p0 = np.load(tmp_path / "params0.npy")
    p1 = np.load(tmp_path / "params1.npy")
    assert np.array_equal(p0, p1), "ranks diverged"
    images, labels, params, perms = _inputs()
    a = _build(torch, params, 2 * CFG["B"], dtype)
    for _ in range(2):
        a.micro_step(images, labels, keep_prob=1.0, perms=perms)
        a.apply_update()
    torch.cuda.synchronize()
    want = a._flat["params"].cpu().numpy()
    err = np.abs(p0 - want).max() / np.abs(want).max()
    assert err < bar, err


@pytest.mark.parametrize("dtype,exchange", [("fp32", {"allreduce_algo": "direct"}), ("bf16", {"grad_dtype": "bf16"})])
def test_prepacked_exchange_is_the_accumulator_path_bit_for_bit(tmp_path, dtype, exchange):
    """accum_num == 1 with the direct exchange: backward packs (g + wd w) / accum_num straight into the exchange's send buffer
    (ops.sgd_accumulate_pack, GradSync(prepacked=True): no accumulate pass, no accumulator, no pack pass) -- the weights after two updates
    are the accumulator path's (config dp_prepack = False), bit for bit, on both ranks."""
    import torch.multiprocessing as mp
    got = {}
    for pre in (True, False):
        d = tmp_path / ("pre%d" % pre)
        d.mkdir()
        mp.spawn(_worker, args=(2, _free_port(), str(d), dtype, dict(exchange, dp_prepack=pre)), nprocs=2, join=True)
        got[pre] = [np.load(d / ("params%d.npy" % r)) for r in range(2)]
        assert np.array_equal(got[pre][0], got[pre][1]), "ranks diverged"
    assert np.array_equal(got[True][0], got[False][0])


def _worker_full_width(rank, world, port, out_dir, tiles, prepack):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from em_adapt_tensorflow_amd.deeplab import ADAPT
    from oracle import deeplab_oracle as do
    from oracle import estep_oracle as eo
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B, size, C = 1, 65, 5
    images, labels = do.synthetic_batch(B * world, size, C, seed=3)
    images = (images / 50.0).astype(np.float32)
    perms = eo.draw_perms(5, C, seed=7)
    a = ADAPT({"input_size": (size, size), "category_num": C, "accum_num": 1, "batch_size": B, "compute_dtype": "bf16", "device": "cuda:0",
               "bf16t": "always", "fused_tiles": tiles, "grad_dtype": "bf16", "dp_prepack": prepack, "force_gradsync": True, "seed": 5,
               "bucket_mb": 32})
    a.build()
    a.pre_train(base_lr=1e-2, weight_decay=1e-3, momentum=0.9, batch_size=B)
    assert a._gradsync is not None and a._gradsync.algo == "direct" and not a._gradsync.writeback and a._gradsync.prepacked == prepack
    sl = slice(rank * B, (rank + 1) * B)
    took = 0
    for _ in range(3):
        a.micro_step(images[sl], labels[sl], keep_prob=0.5, perms=perms)
        a.apply_update()
        took = max(took, len(a._tiles_fresh))
    assert (took > 0) == tiles
    loss = a.eval_loss(images[sl], labels[sl], keep_prob=1.0, perms=perms)
    torch.cuda.synchronize()
    assert float(a._flat["accum"].abs().max()) == 0.0
    imgs = [op.seg[k].view(torch.int16).cpu().numpy() for op in a._plan if op.kind == "conv" for k in ("wt16", "wtd16") if op.seg.get(k) is not None]
    np.savez(os.path.join(out_dir, "full%d.npz" % rank), params=a._flat["params"].cpu().numpy(), momentum=a._flat["momentum"].cpu().numpy(),
             loss=np.float64(loss), **{"img%d" % i: t for i, t in enumerate(imgs)})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("prepack", [True, False])
def test_momentum_on_the_payload_writes_the_tiled_images_the_refresh_would(tmp_path, prepack):
    """The data-parallel job's update (Momentum on the exchange's gathered bf16 payload) on a full-width bf16 net with the LDS-DMA kernel
    forced onto every layer it supports: the layers whose pass also writes their tiled bf16 images (emadapt_sgd_momentum_payload_tiles,
    config fused_tiles) come out with the weights, momentum, images and loss of the path that runs ONE Momentum pass over the flat
    buffer and lets the next forward rebuild the images -- bit for bit, with and without the accumulator (dp_prepack)."""
    import torch.multiprocessing as mp
    got = {}
    for tiles in (True, False):
        d = tmp_path / ("tiles%d" % tiles)
        d.mkdir()
        mp.spawn(_worker_full_width, args=(1, _free_port(), str(d), tiles, prepack), nprocs=1, join=True)
        got[tiles] = dict(np.load(d / "full0.npz"))
    assert set(got[True]) == set(got[False]) and len(got[True]) > 4
    for k in got[True]:
        assert np.array_equal(got[True][k], got[False][k]), k
    assert np.isfinite(float(got[True]["loss"]))


def _worker_status(rank, world, port, out_dir, dtype, exchange):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from em_adapt_tensorflow_amd import ops
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    images, labels, params, perms = _inputs()
    B = CFG["B"]
    a = _build(torch, params, B, dtype, exchange)
    sl = slice(rank * B, (rank + 1) * B)
    a.micro_step(images[sl], labels[sl], keep_prob=1.0, perms=perms)
    a.apply_update()
    torch.cuda.synchronize()
    assert ops.status_poll(clear=False, raise_on_error=False) == 0
    after_one = a._flat["params"].clone()
    mom_one = a._flat["momentum"].clone()
    if rank == 1:
        ops.status_raise(ops.STATUS_STREAMK_TIMEOUT)   # what a stream-K head whose wait ran out leaves behind, on ONE rank
    for _ in range(2):
        a.micro_step(images[sl], labels[sl], keep_prob=1.0, perms=perms)
        a.apply_update()
    torch.cuda.synchronize()
    flags = ops.status_poll(clear=True, raise_on_error=False)
    assert flags & ops.STATUS_NAN_GRADIENT, "rank %d never learnt of the failure (flags 0x%x)" % (rank, flags)
    assert torch.equal(a._flat["params"], after_one) and torch.equal(a._flat["momentum"], mom_one), "rank %d applied an update" % rank
    if a._gradsync.algo != "allreduce":
        assert float(a._flat["accum"].abs().max()) == 0.0
    np.save(os.path.join(out_dir, "params%d.npy" % rank), a._flat["params"].cpu().numpy())
    # after the poll cleared every rank's word the job steps again, together
    a.micro_step(images[sl], labels[sl], keep_prob=1.0, perms=perms)
    a.apply_update()
    torch.cuda.synchronize()
    assert ops.status_poll(clear=True, raise_on_error=False) == 0
    assert not torch.equal(a._flat["params"], after_one) and bool(torch.isfinite(a._flat["params"]).all())
    np.save(os.path.join(out_dir, "params_next%d.npy" % rank), a._flat["params"].cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("dtype,exchange", [("fp32", None), ("fp32", {"allreduce_algo": "direct"}), ("bf16", {"grad_dtype": "bf16"}),
                                            ("fp32", {"allreduce_algo": "direct", "dp_prepack": False})])
def test_a_failure_on_one_rank_stops_every_rank_the_same_way(tmp_path, dtype, exchange):
    """Round-5 advisor finding: only the optimiser kernels looked at the device status word, each at its own.  A rank with a raised word
    skipped its update while its poisoned gradient travelled through the exchange and the clean ranks applied it -- the ranks diverged and
    rank 0 could save NaN weights.  Now (emadapt_status_collective) the rank's contribution to the exchange is NaN while its word is
    raised, every rank's Momentum pass leaves NaN elements alone and raises its OWN word: no rank moves, every rank's poll reports, and
    once the words are cleared the ranks step on together, bit-identical.  Ring all-reduce, the direct exchange with and without the
    accumulator, and the bf16 payload."""
    import torch.multiprocessing as mp
    mp.spawn(_worker_status, args=(2, _free_port(), str(tmp_path), dtype, exchange), nprocs=2, join=True)
    for name in ("params", "params_next"):
        p0, p1 = np.load(tmp_path / (name + "0.npy")), np.load(tmp_path / (name + "1.npy"))
        assert np.array_equal(p0, p1) and np.isfinite(p0).all(), name
