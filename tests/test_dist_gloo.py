"""World-size-2 `gloo` tests (CPU) of the data-parallel path: GradSync buckets over the flat accumulator
buffer + the sum -> 1/world -> Momentum semantics that reproduce the reference's ``accum += g/accum_num``
(deeplab.py:202) with micro-batches mapped to ranks.  No GPU, no kernels: the exchange logic only."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from em_adapt_tensorflow_amd.dist import GradSync, all_reduce_max

    # a flat accumulator with three "layers" laid out in backward order, gaps from 64-float alignment
    segs = [(0, 1088), (1088, 1088 + 4160), (5248, 5248 + 320)]
    total = segs[-1][1]
    rng = np.random.default_rng(100 + rank)
    flat = torch.tensor(rng.normal(0, 1, total).astype(np.float32))
    mine = flat.clone()
    gs = GradSync(flat, segs, bucket_bytes=2048 * 4)  # several buckets, boundaries inside segments
    assert gs.world_size == world and len(gs.buckets) == 3
    assert gs.timing is False and gs._ev.maxlen is not None   # no events unless asked for, and a bounded record when asked
    # segments must be reported in layout (= backward) order
    with pytest.raises(RuntimeError):
        gs.segment_ready(1088, 5248)
    launched = []
    for s, e in segs:
        gs.segment_ready(s, e)
        launched.append(gs._next_bucket)
    assert launched == [0, 2, 3], launched  # bucket k goes out as soon as the watermark passes its end
    gs.wait()
    assert gs.launched_bytes == total * 4
    # second round re-uses the object (state reset by wait); this time the exchange is held back until the last layer
    # (an opt-in policy of the training step: dist.py hold / release, config "allreduce_release_at")
    flat2_before = flat.clone()
    gs.hold()
    gs.segment_ready(*segs[0])
    gs.segment_ready(*segs[1])
    assert gs._next_bucket == 0, "held buckets must not be launched"
    gs.release()
    assert gs._next_bucket == 2
    gs.segment_ready(*segs[2])
    assert gs._next_bucket == 3
    gs.wait()
    np.save(os.path.join(out_dir, "mine%d.npy" % rank), mine.numpy())
    np.save(os.path.join(out_dir, "sum%d.npy" % rank), flat2_before.numpy())
    np.save(os.path.join(out_dir, "sum2_%d.npy" % rank), flat.numpy())
    # wait() before all segments are ready is an error
    gs.segment_ready(*segs[0])
    with pytest.raises(RuntimeError):
        gs.wait()
    m = all_reduce_max(10.0 + rank, torch.device("cpu"))
    assert m == 10.0 + world - 1
    dist.barrier()
    dist.destroy_process_group()


def test_gradsync_world2(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    mine = [np.load(tmp_path / ("mine%d.npy" % r)) for r in range(world)]
    want = mine[0] + mine[1]
    for r in range(world):
        got = np.load(tmp_path / ("sum%d.npy" % r))
        assert np.array_equal(got, want), "rank %d: all-reduce(sum) of the accumulators is wrong" % r
        assert np.array_equal(np.load(tmp_path / ("sum2_%d.npy" % r)), want * 2)


def _dp_equivalence_worker(rank, world, port, out_dir):
    """Each rank owns half of a global batch.  Rank-local mean gradients, summed and scaled by 1/world inside
    the Momentum step, must equal the single-process gradient of the global-batch mean loss."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from em_adapt_tensorflow_amd.dist import GradSync
    from oracle import deeplab_oracle as do
    from oracle import estep_oracle as eo

    C, size, B = 5, 33, 2
    oracle = do.DeepLabOracle(category_num=C, width=4, dtype=torch.float64)
    params = do.init_params(oracle.specs, seed=4, he=True)
    images, labels = do.synthetic_batch(world * B, size, C, seed=8)
    images = images / 50
    perms = eo.draw_perms(5, C, seed=1)
    sl = slice(rank * B, (rank + 1) * B)
    r = oracle.micro_step(params, images[sl], labels[sl], perms, weight_decay=1e-3)
    names = list(reversed(list(params.keys())))  # backward order: fc8 first
    chunks, segs, off = [], [], 0
    for n in names:
        for g in r["grads"][n]:
            chunks.append(torch.tensor(g.reshape(-1)))
        n_el = sum(g.size for g in r["grads"][n])
        segs.append((off, off + n_el))
        off += n_el
    flat = torch.cat(chunks).to(torch.float64)
    gs = GradSync(flat, segs, bucket_bytes=4096)
    for s, e in segs:
        gs.segment_ready(s, e)
    gs.wait()
    flat /= world
    if rank == 0:
        full = oracle.micro_step(params, images, labels, perms, weight_decay=1e-3)
        want = torch.cat([torch.tensor(g.reshape(-1)) for n in names for g in full["grads"][n]]).to(torch.float64)
        err = float((flat - want).abs().max() / want.abs().max())
        with open(os.path.join(out_dir, "err.txt"), "w") as f:
            f.write("%g" % err)
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_equals_global_batch(tmp_path):
    port = _free_port()
    mp.spawn(_dp_equivalence_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    err = float(open(tmp_path / "err.txt").read())
    assert err < 1e-9, err
