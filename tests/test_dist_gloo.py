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


def _direct_worker(rank, world, port, out_dir):
    """algo="direct" (reduce-scatter + all-gather as two all-to-all steps, fp32 accumulation by the shard's owner, in rank order)
    with the fp32 and the bf16 payload; bucket sizes that are not multiples of the world size (padding lanes)."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from em_adapt_tensorflow_amd.dist import GradSync

    segs = [(0, 1088), (1088, 1088 + 4163), (5251, 5251 + 321)]
    total = segs[-1][1]
    rng = np.random.default_rng(200 + rank)
    mine = torch.tensor((rng.normal(0, 1, total) * np.exp(rng.normal(0, 3, total))).astype(np.float32))   # wide dynamic range
    with pytest.raises(ValueError):
        GradSync(mine.clone(), segs, algo="allreduce", payload="bf16")   # a bf16 ring would round the running sum at every hop
    for payload in ("fp32", "bf16"):
        flat = mine.clone()
        gs = GradSync(flat, segs, bucket_bytes=2048 * 4, algo="direct", payload=payload)
        assert all((e - s) % world == 0 for s, e in gs.buckets[:-1])
        el = 2 if payload == "bf16" else 4
        assert gs.wire_bytes_per_update == int(2 * (world - 1) / world * total * el)
        gs.hold()
        gs.segment_ready(*segs[0])
        assert gs._next_bucket == 0
        gs.release()
        for s, e in segs[1:]:
            gs.segment_ready(s, e)
        gs.wait()
        assert gs.launched_bytes == total * 4 and gs.stats()["payload"] == payload
        np.save(os.path.join(out_dir, "%s_%d.npy" % (payload, rank)), flat.numpy())
        flat.copy_(mine)                     # a second update re-uses the object and its staging buffers
        for s, e in segs:
            gs.segment_ready(s, e)
        gs.wait()
        assert np.array_equal(flat.numpy(), np.load(os.path.join(out_dir, "%s_%d.npy" % (payload, rank))))
        # writeback=False (what ADAPT uses: the optimiser kernel reads the gathered payload in place): flat keeps the LOCAL
        # values, `gathered` holds the sum in the payload's type, every bucket at its own offset
        flat.copy_(mine)
        gs2 = GradSync(flat, segs, bucket_bytes=2048 * 4, algo="direct", payload=payload, writeback=False)
        assert gs2.gathered.dtype == (torch.bfloat16 if payload == "bf16" else torch.float32) and gs2.gathered.numel() >= total
        for s, e in segs:
            gs2.segment_ready(s, e)
        gs2.wait()
        assert np.array_equal(flat.numpy(), mine.numpy())
        assert np.array_equal(gs2.gathered[:total].float().numpy(), np.load(os.path.join(out_dir, "%s_%d.npy" % (payload, rank))))
        # prepacked (accum_num == 1: backward writes this rank's contribution, already in the payload type, into `sendfull`; the
        # accumulator is never read): the same sums, bit for bit -- flat may hold anything
        flat.fill_(float("nan"))
        gs3 = GradSync(flat, segs, bucket_bytes=2048 * 4, algo="direct", payload=payload, writeback=False, prepacked=True)
        assert gs3.prepacked and gs3.sendfull.dtype == gs3.gathered.dtype and gs3.sendfull.numel() == gs3.gathered.numel()
        for s, e in segs:
            gs3.sendfull[s:e].copy_(mine[s:e])          # (the HIP path: ops.sgd_accumulate_pack writes it)
            gs3.segment_ready(s, e)
        gs3.wait()
        assert np.array_equal(gs3.gathered[:total].float().numpy(), np.load(os.path.join(out_dir, "%s_%d.npy" % (payload, rank))))
        assert not GradSync(flat, segs, algo="allreduce", prepacked=True).prepacked      # (only the direct exchange without write-back)
        # explicit bucket boundaries (round 6: ADAPT cuts the direct exchange behind the last fully connected layer): an offset is rounded
        # down to whole shards, the first bucket leaves as soon as the segment it names is ready, the sums are the same bits
        flat.copy_(mine)
        gs4 = GradSync(flat, segs, algo="direct", payload=payload, writeback=False, boundaries=[segs[1][1]])
        cut = segs[1][1] // world * world
        assert gs4.buckets == [(0, cut), (cut, total)]
        gs4.segment_ready(*segs[0])
        assert gs4._next_bucket == 0
        gs4.segment_ready(*segs[1])
        assert gs4._next_bucket == 1                    # (77 % of the bytes are on their way before the last segment exists)
        gs4.segment_ready(*segs[2])
        gs4.wait()
        assert np.array_equal(gs4.gathered[:total].float().numpy(), np.load(os.path.join(out_dir, "%s_%d.npy" % (payload, rank))))
    np.save(os.path.join(out_dir, "mine%d.npy" % rank), mine.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])   # (8: the node the scaling bench runs on; the last bucket is 4 elements short of whole shards)
def test_direct_exchange_and_bf16_payload(tmp_path, world):
    port = _free_port()
    mp.spawn(_direct_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    mine = [torch.tensor(np.load(tmp_path / ("mine%d.npy" % r))) for r in range(world)]
    # fp32 payload: contributions added in rank order, in fp32, by one owner -- every rank holds the same bits
    want32 = mine[0].clone()
    for m in mine[1:]:
        want32 += m
    # bf16 payload: bf16(sum_r fp32(bf16(g_r))), the sum in fp32 in rank order
    acc = mine[0].bfloat16().float()
    for m in mine[1:]:
        acc += m.bfloat16().float()
    want16 = acc.bfloat16().float()
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / ("fp32_%d.npy" % r)), want32.numpy()), r
        assert np.array_equal(np.load(tmp_path / ("bf16_%d.npy" % r)), want16.numpy()), r      # exact, hence identical on all ranks
    # and the bf16 exchange equals the fp32 exchange to bf16 rounding: one rounding per contribution + one of the sum
    bound = sum(m.abs() for m in mine) * 2.0 ** -8 + want32.abs() * 2.0 ** -8
    assert bool(((want16 - want32).abs() <= bound).all())
    assert float((want16 - want32).abs().max()) > 0            # (it IS a different number)
