"""CPU tests of the host-side mirror of the reference interface: plan construction, parameter shapes
(deeplab.py:131-149), flat-buffer layout, FLOP/parameter counts of SURVEY section 8, perms drawing."""
import os
import random

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _adapt(**kw):
    from em_adapt_tensorflow_amd.deeplab import ADAPT
    cfg = {"input_size": (321, 321), "category_num": 21, "device": "cpu"}
    cfg.update(kw)
    a = ADAPT(cfg)
    a.build()
    return a


def test_needs_gpu_unless_device_given():
    from em_adapt_tensorflow_amd.deeplab import ADAPT
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        ADAPT({"input_size": (321, 321)})


def test_reference_shapes_and_counts():
    a = _adapt(width=64)
    shapes = {k: tuple(w.shape) for k, (w, b) in a.weights.items()}
    assert shapes["conv1_1"] == (3, 3, 3, 64) and shapes["conv2_1"] == (3, 3, 64, 128)
    assert shapes["conv4_1"] == (3, 3, 256, 512) and shapes["conv5_3"] == (3, 3, 512, 512)
    assert shapes["fc6"] == (4, 4, 512, 4096) and shapes["fc7"] == (1, 1, 4096, 4096) and shapes["fc8"] == (1, 1, 4096, 21)
    assert a.parameter_count() == 65_140_565                      # 65.14 M
    fwd, fb = a.flops_per_image()
    assert abs(fwd / 1e9 - 251.9) < 0.1 and abs(fb / 1e9 - 755.7) < 0.2
    assert a.out_hw == (41, 41)
    names = [op.out_name for op in a._plan]
    assert names[:3] == ["relu1_1", "relu1_2", "pool1"] and names[-3:] == ["drop6", "drop7", "fc8"]
    rates = {op.name: op.rate for op in a._plan if op.kind == "conv"}
    assert rates["conv5_2"] == 2 and rates["fc6"] == 4 and rates["conv4_3"] == 1
    assert [op.stride for op in a._plan if op.kind == "pool"] == [2, 2, 2, 1, 1]
    assert len(a.trainable_list) == 32 and a.trainable_list[0] is a.weights["conv1_1"][0]


def test_largefov_and_513():
    a = _adapt(preset="largefov", input_size=(513, 513), category_num=41)
    assert tuple(a.weights["fc6"][0].shape) == (3, 3, 512, 1024)
    assert a.out_hw == (65, 65)
    fwd, fb = a.flops_per_image()
    assert abs(fb / 1e9 - 772.0) < 0.5
    assert {op.name: op.rate for op in a._plan if op.kind == "conv"}["fc6"] == 12


def test_flat_layout_is_backward_ordered_and_padded():
    a = _adapt(width=8, input_size=(65, 65))
    segs = a._segments
    order = sorted(segs, key=lambda k: segs[k]["start"])
    assert order[0] == "fc8" and order[-1] == "conv1_1"
    for k, s in segs.items():
        assert s["start"] % 64 == 0 and s["nw"] % 4 == 0 and s["nb"] % 4 == 0
    assert segs["conv1_1"]["pshape"] == (3, 3, 4, 8) and segs["fc8"]["pshape"][3] == 24
    w = a.weights["conv1_1"][0]
    assert w.data_ptr() == a._flat["params"][segs["conv1_1"]["start"]:].data_ptr()
    # padding lanes are zero after init; the weights themselves are ~N(0, 0.01^2)
    wp = segs["conv1_1"]["w_pad"]
    assert float(wp[:, :, 3, :].abs().max()) == 0.0
    assert 0.005 < float(a.weights["fc6"][0].std()) < 0.02 and float(a.weights["fc6"][1].abs().max()) == 0.0
    # views write through to the flat buffer
    a.weights["fc8"][1].fill_(3.0)
    s = segs["fc8"]
    assert float(a._flat["params"][s["start"] + s["nw"]]) == 3.0


def test_default_perms_follow_python_random_like_the_reference():
    from em_adapt_tensorflow_amd.estep import draw_perms
    random.seed(5)
    got = draw_perms(5, 21)
    r = random.Random(5)
    want = []
    for _ in range(5):
        t = list(range(1, 21))
        r.shuffle(t)
        want.append(t)
    assert np.array_equal(got, np.array(want))


def test_synthetic_data_shapes():
    from em_adapt_tensorflow_amd.data import SyntheticData
    d = SyntheticData({"input_size": (33, 33), "category_num": 21, "device": "cpu", "length": 10})
    it = d.next_batch("train", 3)
    x, y = next(it)
    assert tuple(x.shape) == (3, 33, 33, 3) and tuple(y.shape) == (3, 33, 33, 1)
    vals = set(np.unique(y.numpy()).tolist())
    assert 0.0 in vals and vals <= set(float(v) for v in range(21)) | {255.0}
    assert d.get_data_len() == 10


def test_launch_plans_at_the_benchmark_shapes():
    """The plans the batch-16 layers get (csrc/conv_igemm.hip plan_launches; no device needed): fc6 / fc7 forward run
    256x256 tiles on the rows that fill whole rounds of the 256 CUs and a split 128x128 launch on the rest."""
    from em_adapt_tensorflow_amd import ops
    M = 16 * 41 * 41
    assert ops.conv_plan(16, 41, 41, 512, 4096, 4, 4)[:2] == (5, M)                # fc6 fwd: stream-K over row bands (padding rows
    with ops.tuned(fp32_loader=2):                                                 #   of the dilated filter skipped); without them:
        assert ops.conv_plan(16, 41, 41, 512, 4096, 4, 4)[:3] == (4, 24576, 1)     #   96 x 16 tiles = 6 rounds + a tail launch
    assert ops.conv_plan(16, 41, 41, 4096, 4096, 1, 1)[:2] == (4, 24576)           # fc7 fwd
    assert ops.conv_plan(16, 41, 41, 4096, 4096, 1, 1, dgrad=True)[:2] == (4, 24576)
    assert ops.conv_plan(16, 41, 41, 4096, 24, 1, 1, dgrad=True)[0] == 1           # fc8 dgrad (N = 4096, K = 24: one K step -> small tiles)
    p = ops.conv_plan(16, 41, 41, 512, 4096, 4, 4, dgrad=True)                     # fc6 dgrad (N = 512: 212 big tiles < 256 CUs,
    assert p[0] == 5 and p[1] == M and p[2] >= 2 and p[3] == 25                    #   K = 65536): stream-K over 5 x 5 pixel classes
    for batch in (2, 6, 8, 32):   # packed rows pay at every batch size (round 2's whole-tile classes: not below 12); the
        assert ops.conv_plan(batch, 41, 41, 512, 4096, 4, 4)[::3] == (5, 25)       # reference's own batch is 6 (deeplab.py:288)
    assert ops.conv_plan(10, 65, 65, 512, 4096, 4, 4)[::3] == (5, 25)              # BASELINE configs[4]: 65 x 65 maps
    assert ops.conv_plan(16, 41, 41, 512, 512, 3, 1) == (5, M, 3, 0)               # conv4_x: stream-K, a tile has <= 3 pieces
    assert ops.conv_plan(16, 41, 41, 512, 512, 3, 2)[3] == 9                       # conv5_x (rate 2): 3 x 3 classes, 6.2 % of the products
    assert ops.conv_plan(16, 41, 41, 512, 512, 3, 1)[3] == 0                       # conv4_x (rate 1): 2.7 % -- below the threshold
    with ops.tuned(pack_pct=2):
        assert ops.conv_plan(16, 41, 41, 512, 512, 3, 1)[3] == 9
    assert ops.conv_plan(16, 41, 41, 512, 1024, 3, 12)[3] == 9                     # LargeFOV fc6 (rate 12): 3 x 3 pixel classes
    assert ops.conv_plan(16, 81, 81, 256, 256, 3, 1)[:2] == (5, 16 * 81 * 81)      # conv3_x: 411 tiles = 1.6 rounds -> stream-K
    with ops.tuned(plan_cus=137):                                                  # 411 = 3 x 137 tiles: whole rounds, no split at all
        assert ops.conv_plan(16, 81, 81, 256, 256, 3, 1) == (4, 16 * 81 * 81, 1, 0)
    assert ops.conv_plan(16, 161, 161, 128, 128, 3, 1)[0] == 3                     # conv2_2: N = 128
    assert ops.conv_plan(16, 321, 321, 64, 64, 3, 1)[0] == 1                       # conv1_2: N <= 64
    with ops.tuned(fp32_tile=2, plan_cus=4):
        assert ops.conv_plan(2, 23, 29, 40, 72, 3, 1)[:2] == (2, 1024)             # forced: 4 of 6 row tiles + a tail
    assert ops.tune_get("fp32_tile") == 0
    from em_adapt_tensorflow_amd import _lib
    with pytest.raises(_lib.EmadaptError):
        ops.tune_set("no_such_knob", 1)


def _list_only_dataset(tmp_path, n, **cfg):
    from em_adapt_tensorflow_amd.data import dataset
    (tmp_path / "txt").mkdir(exist_ok=True)
    ids = ["img%03d" % i for i in range(n)]
    for cat in ("train", "val"):
        (tmp_path / "txt" / (cat + ".txt")).write_text("\n".join(ids) + "\n")
    base = {"input_size": (33, 33), "main_path": str(tmp_path), "list_path": str(tmp_path / "txt"), "device": "cpu"}
    base.update(cfg)
    return dataset(base)


def test_dataset_shards_every_global_batch_across_ranks(tmp_path):
    """Data-parallel input: with a common seed all ranks draw the same permutation and take disjoint slices of every
    global batch; the ragged remainder of an epoch is dropped so shards stay equal-sized."""
    world, bs, n = 2, 3, 20
    orders = []
    for rank in range(world):
        ds = _list_only_dataset(tmp_path, n, seed=11, rank=rank, world_size=world)
        assert (ds.rank, ds.world) == (rank, world)
        orders.append(list(ds.sample_order("train", bs, epoches=2)))
    assert len(orders[0]) == len(orders[1]) == 2 * (n // (bs * world))
    for (e0, s0, i0), (e1, s1, i1) in zip(*orders):
        assert (e0, s0) == (e1, s1) and len(i0) == len(i1) == bs
        assert not set(i0.tolist()) & set(i1.tolist()), "ranks must not see the same sample in one step"
    for ep in range(2):
        seen = [i for o in orders for e, s, idx in o if e == ep for i in idx.tolist()]
        assert len(seen) == len(set(seen)) == (n // (bs * world)) * bs * world    # each sample at most once per epoch
    # augmentation draws differ per rank, single-process behaviour is unchanged (short last batch kept)
    a = _list_only_dataset(tmp_path, n, seed=11, rank=0, world_size=2).rng.uniform()
    b = _list_only_dataset(tmp_path, n, seed=11, rank=1, world_size=2).rng.uniform()
    assert a != b
    single = list(_list_only_dataset(tmp_path, 5, seed=1).sample_order("train", 3, epoches=1))
    assert [len(i) for _, _, i in single] == [3, 2]


def test_dataset_epoch_order_is_stateless_and_small_sets_fail_loudly(tmp_path):
    """A rank-local pass (rank 0 evaluating on "val", say) must not shift that rank's later "train" shards against the
    other ranks': the permutation of (category, epoch) is a function of the seed alone.  A set smaller than one global
    batch is an error at the first draw, not a bare StopIteration out of the training loop."""
    world, bs, n = 2, 3, 20
    r0 = _list_only_dataset(tmp_path, n, seed=5, rank=0, world_size=world)
    r1 = _list_only_dataset(tmp_path, n, seed=5, rank=1, world_size=world)
    list(r0.sample_order("val", bs, epoches=3))                      # only rank 0 walks "val"
    o0 = list(r0.sample_order("train", bs, epoches=2))
    o1 = list(r1.sample_order("train", bs, epoches=2))
    for (_, _, i0), (_, _, i1) in zip(o0, o1):
        assert not set(i0.tolist()) & set(i1.tolist())
    for ep in range(2):
        seen = [i for o in (o0, o1) for e, s, idx in o if e == ep for i in idx.tolist()]
        assert len(seen) == len(set(seen))
    assert np.array_equal(r0.epoch_perm("train", 1), r1.epoch_perm("train", 1))
    assert not np.array_equal(r0.epoch_perm("train", 0), r0.epoch_perm("train", 1))
    assert not np.array_equal(r0.epoch_perm("train", 0), r0.epoch_perm("val", 0))
    small = _list_only_dataset(tmp_path, 5, seed=1, rank=0, world_size=2)
    with pytest.raises(ValueError, match="fewer than one global batch"):
        next(small.sample_order("train", 3))


def test_resumed_runs_continue_in_their_epoch(tmp_path):
    """A resumed run (config "start_epoch") or a second pass over the same data object must not replay epoch 0, 1, ...: the order
    of an epoch is a function of (seed, category, epoch), and sample_order starts where it is told to (advisor finding, round 3)."""
    ds = _list_only_dataset(tmp_path, 12, seed=3)
    full = list(ds.sample_order("train", 4, epoches=4))
    tail = list(ds.sample_order("train", 4, epoches=2, start_epoch=2))
    assert [e for e, _, _ in tail] == [2, 2, 2, 3, 3, 3]
    for (e0, s0, i0), (e1, s1, i1) in zip(full[6:], tail):
        assert (e0, s0) == (e1, s1) and np.array_equal(i0, i1)
    assert not np.array_equal(full[0][2], tail[0][2])                 # (epoch 2 is not epoch 0)
    # next_batch without start_epoch continues behind what this object has handed out (decode / preprocess stubbed out)
    class FakePipe:
        def submit(self, k, paths): return (k, len(paths))
        def collect(self, ticket): return [(8, 8)] * ticket[1]
        def preprocess(self, ticket, samples): return torch.zeros(len(samples)), torch.zeros(len(samples))
        def close(self): pass
    ds._make_pipeline = lambda batch_size, slots: FakePipe()
    first = ds.next_batch("train", 4, epoches=1)
    ids_a = [ids for _, _, ids in first]
    assert ds.info["next_epoch"]["train"] == 1
    second = ds.next_batch("train", 4, epoches=1)
    ids_b = [ids for _, _, ids in second]
    assert ds.info["epoch"]["train"] == 1 and ds.info["next_epoch"]["train"] == 2
    assert sorted(sum(ids_a, [])) == sorted(sum(ids_b, [])) and ids_a != ids_b     # same samples, epoch 1's order
    again = [ids for _, _, ids in ds.next_batch("train", 4, epoches=1, start_epoch=0)]
    assert again == ids_a


def test_decode_worker_process_fills_the_shared_staging_block(tmp_path):
    """f1, host side (data._Pipeline, decode_mode "process"): a decode worker is a separate PROCESS (python -m ..._decode <shm>, stdlib +
    NumPy + PIL only) that answers one line per request and writes the decoded bytes into the shared-memory block at the offsets it
    is told: uint8 RGB rows at img_off, the palette indices at lab_off.  Several workers, requests in flight, a file that cannot be
    decoded, a sample larger than its slot; the worker ends when its stdin closes."""
    import subprocess
    import sys
    from multiprocessing import shared_memory
    from PIL import Image
    rng = np.random.default_rng(3)
    files = []
    for i, (h, w) in enumerate([(37, 53), (64, 48), (50, 50), (41, 77)]):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        lab = rng.integers(0, 21, (h, w), dtype=np.uint8)
        lab[:2] = 255
        Image.fromarray(img).save(tmp_path / ("%d.jpg" % i), quality=95)
        pl = Image.fromarray(lab, mode="P")
        pl.putpalette([c for k in range(256) for c in ((k * 37) % 256, (k * 91) % 256, (k * 53) % 256)])
        pl.save(tmp_path / ("%d.png" % i))
        files.append((str(tmp_path / ("%d.jpg" % i)), str(tmp_path / ("%d.png" % i)), lab))
    mp_, n = 80 * 80, len(files)
    cap = 4 * mp_
    shm = shared_memory.SharedMemory(create=True, size=cap * n)
    procs = []
    try:
        host = np.ndarray((cap * n,), dtype=np.uint8, buffer=shm.buf)
        host[:] = 7
        env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
        procs = [subprocess.Popen([sys.executable, "-m", "em_adapt_tensorflow_amd._decode", shm.name], env=env, stdin=subprocess.PIPE,
                                  stdout=subprocess.PIPE, text=True, bufsize=1) for _ in range(2)]
        for j, (jf, pf, _) in enumerate(files):               # all requests first: two per worker in flight
            procs[j % 2].stdin.write("%d\t%d\t%d\t%s\t%s\n" % (j * cap, j * cap + 3 * mp_, cap, jf, pf))
        for p in procs:
            p.stdin.flush()
        for j, (jf, pf, lab) in enumerate(files):
            reply = procs[j % 2].stdout.readline().split()
            assert reply[0] == "OK", reply
            h, w = int(reply[1]), int(reply[2])
            assert (h, w) == lab.shape
            want = np.asarray(Image.open(jf).convert("RGB"))
            assert np.array_equal(host[j * cap:j * cap + h * w * 3].reshape(h, w, 3), want)
            assert np.array_equal(host[j * cap + 3 * mp_:j * cap + 3 * mp_ + h * w].reshape(h, w), lab)
            assert host[j * cap + h * w * 3] == 7               # nothing behind the sample was touched
        (tmp_path / "bad.jpg").write_bytes(b"not a jpeg")
        procs[0].stdin.write("0\t%d\t%d\t%s\t%s\n" % (3 * mp_, cap, tmp_path / "bad.jpg", files[0][1]))
        procs[0].stdin.write("0\t%d\t%d\t%s\t%s\n" % (3 * 100, 4 * 100, files[0][0], files[0][1]))     # a slot of 100 pixels
        procs[0].stdin.flush()
        r1, r2 = procs[0].stdout.readline(), procs[0].stdout.readline()
        assert r1.startswith("ERR ") and r2.startswith("ERR ") and "larger than the staging slot" in r2
        for p in procs:
            p.stdin.close()
        assert [p.wait(timeout=20) for p in procs] == [0, 0]
        del host
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        shm.close()
        shm.unlink()


def test_synthetic_data_differs_per_rank():
    from em_adapt_tensorflow_amd.data import SyntheticData
    cfg = {"input_size": (33, 33), "category_num": 21, "device": "cpu", "seed": 0}
    y0 = next(SyntheticData(dict(cfg, rank=0, world_size=2)).next_batch("train", 2))[0]
    y1 = next(SyntheticData(dict(cfg, rank=1, world_size=2)).next_batch("train", 2))[0]
    assert not torch.equal(y0, y1)


def test_latest_checkpoint_is_chosen_by_step_number_or_index(tmp_path):
    """network.py:104-110 with checkpoint=True resumes from the LAST checkpoint written (tf.train.get_checkpoint_state),
    not the lexicographically largest file name ("norm-5999" sorts after "norm-11999")."""
    import json
    from em_adapt_tensorflow_amd.network import latest_checkpoint
    for name in ("norm-5999.npz", "norm-11999.npz", "lr-0.001000-100.npz"):
        (tmp_path / name).write_bytes(b"x")
    assert latest_checkpoint(str(tmp_path)).endswith("norm-11999.npz")
    (tmp_path / "checkpoint").write_text(json.dumps({"model_checkpoint_path": "norm-5999.npz"}))
    assert latest_checkpoint(str(tmp_path)).endswith("norm-5999.npz")      # the index file wins: it records what was written last
    (tmp_path / "empty").mkdir()
    with pytest.raises(FileNotFoundError):
        latest_checkpoint(str(tmp_path / "empty"))


def test_stream_k_plans_and_their_workspace_are_consistent():
    """Launch plans need no device: over a sweep of shapes and CU counts the stream-K plans (variant 5) report the largest
    number of workgroups that share a tile, and the workspace query asks for exactly the arrival words + one 256 KB accumulator
    image per workgroup (tiles are combined inside the launch: csrc/igemm_v2.h) when any tile is shared, plus the row table."""
    import ctypes
    from em_adapt_tensorflow_amd import _lib, ops
    lib = _lib.load()
    lib.emadapt_conv_fwd_workspace_bytes.restype = ctypes.c_size_t
    lib.emadapt_conv_dgrad_workspace_bytes.restype = ctypes.c_size_t
    rng = np.random.default_rng(0)
    seen = set()
    for _ in range(300):
        B, H, W = int(rng.integers(1, 17)), int(rng.integers(5, 90)), int(rng.integers(5, 90))
        Cin, Cout = int(rng.choice([16, 32, 64, 128, 256, 512, 1024])), int(rng.choice([24, 64, 128, 256, 512, 1040, 4096]))
        k, dil = int(rng.choice([1, 3, 4])), int(rng.choice([1, 2, 4]))
        cus = int(rng.choice([0, 3, 7, 64, 256]))
        M = B * H * W
        with ops.tuned(plan_cus=cus):
            for dgrad in (False, True):
                cfg, main_rows, main_s, tail_s = ops.conv_plan(B, H, W, Cin, Cout, k, dil, dgrad=dgrad)
                N = Cin if dgrad else Cout
                d = ops.conv_desc(B, H, W, Cin, Cout, k, k, dil)
                ws = (lib.emadapt_conv_dgrad_workspace_bytes if dgrad else lib.emadapt_conv_fwd_workspace_bytes)(ctypes.byref(d))
                assert 1 <= cfg <= 5 and 0 < main_rows <= M and main_s >= 0 and tail_s >= 0
                seen.add(cfg)
                if cfg == 5:   # (plan[3] = number of pixel classes of a padding-skipping launch, 0 or >= 2; such a launch also
                    #  keeps its GEMM row -> pixel table, one int per row of its whole tiles, behind the slab levels)
                    table = -(-M // 256) * 256 * 4 if tail_s else 0
                    g = cus or 256
                    sk = -(-g * 4 // 256) * 256 + g * 256 * 256 * 4
                    assert main_rows == M and tail_s != 1 and tail_s <= 36 and ws == (sk if main_s else 0) + table, (cfg, main_rows, main_s, tail_s, ws)
                    assert main_s != 1, "a tile with partial sums has at least two pieces"
                elif main_rows < M:
                    g = cus or 256
                    sk = -(-g * 4 // 256) * 256 + g * 256 * 256 * 4
                    assert cfg in (2, 4) and main_rows % 256 == 0
                    assert ws >= min(sk, (M - main_rows) * N * 4 * tail_s) * (1 if tail_s > 1 else 0)
                else:
                    assert ws == (main_s * M * N * 4 if main_s > 1 else 0)
    assert {1, 3, 4, 5} <= seen, seen


def test_plane_kernel_plans_sizes_and_shape_rules_need_no_device():
    """Host side of the bf16x6 plane-image kernels (csrc/conv_x6p.hip): which convolutions they take, their launch plans with
    the workspace each asks for, and the sizes of the plane / tiled filter images."""
    import ctypes
    from em_adapt_tensorflow_amd import _lib, ops
    lib = _lib.load()
    # plane image: 6 bytes per element; tiled filter image: rows padded to whole 256-row tiles, reduction in groups of 16
    assert lib.emadapt_planes_bytes(26896, 512) == 26896 * 512 * 6
    assert ops.filter_tiles_bytes(512, 4096, 4, 4) == 16 * (512 // 16) * 4096 * 96            # fc6 forward: rows = Cout
    assert ops.filter_tiles_bytes(512, 4096, 4, 4, dgrad=True) == 16 * (4096 // 16) * 512 * 96  # data gradient: rows = Cin
    assert ops.filter_tiles_bytes(4096, 24, 1, 1) == (4096 // 16) * 256 * 96                  # fc8: 24 rows in one padded tile
    assert ops.filter_tiles_bytes(4096, 24, 1, 1, dgrad=True) == 0                            # k = 24 is not a whole group of 16
    # which layers of the benchmark network (batch 16, 321 x 321) run on the plane kernels: (forward, data gradient, filter gradient)
    taken = lambda *s: tuple(ops.conv_x6p_supported(ops.conv_desc(*s), kind) for kind in (0, 1, 2))
    assert taken(16, 321, 321, 4, 64, 3, 3, 1) == (False, False, False)     # conv1_1
    assert taken(16, 161, 161, 128, 128, 3, 3, 1) == (False, False, False)  # conv2_2: 128 output columns stay on conv_x6.hip
    assert taken(16, 81, 81, 128, 256, 3, 3, 1) == (True, False, False)     # conv3_1: dgrad has 128 columns, wgrad needs Cin % 256 == 0
    assert taken(16, 81, 81, 256, 256, 3, 3, 1) == (True, True, True)       # conv3_2
    assert taken(16, 41, 41, 512, 512, 3, 3, 2) == (True, True, True)       # conv5_x
    assert taken(16, 41, 41, 512, 4096, 4, 4, 4) == (True, True, True)      # fc6
    assert taken(16, 41, 41, 4096, 4096, 1, 1, 1) == (True, True, True)     # fc7
    assert taken(16, 41, 41, 4096, 24, 1, 1, 1) == (False, False, False)    # fc8
    # ... and where they have enough work to fill 256 CUs (a host with both kernel families asks this: one image's conv4_x is
    # 14 tiles of 288 K steps -- the split-K launches of conv_x6.hip are the better choice there)
    pays = lambda *s: tuple(ops.conv_x6p_pays(ops.conv_desc(*s), kind) for kind in (0, 1, 2))
    assert pays(16, 41, 41, 512, 512, 3, 3, 1) == (True, True, True) and pays(1, 41, 41, 512, 512, 3, 3, 1) == (False, False, True)
    assert pays(1, 41, 41, 4096, 4096, 1, 1, 1) == (True, True, True) and pays(1, 81, 81, 256, 256, 3, 3, 1)[:2] == (False, False)
    assert pays(16, 41, 41, 4096, 24, 1, 1, 1) == (False, False, False)
    M = 16 * 41 * 41
    # plans: (rows of the first launch, stream-K?, its slab levels, pixel classes of packed rows, tail slab levels / -1 plain)
    assert ops.conv_x6p_plan(ops.conv_desc(16, 41, 41, 4096, 4096, 1, 1, 1)) == (24576, 0, 0, 0, 3)   # fc7: 6 whole rounds + stream-K tail
    assert ops.conv_x6p_plan(ops.conv_desc(16, 41, 41, 512, 4096, 4, 4, 4))[:2] == (M, 1)             # fc6: stream-K over packed rows
    assert ops.conv_x6p_plan(ops.conv_desc(16, 41, 41, 512, 4096, 4, 4, 4))[3] == 25
    assert ops.conv_x6p_plan(ops.conv_desc(16, 41, 41, 512, 512, 3, 3, 2))[3] == 9                    # conv5_x: 3 x 3 pixel classes
    assert ops.conv_x6p_plan(ops.conv_desc(16, 41, 41, 512, 512, 3, 3, 1)) == (M, 1, 3, 0, 0)         # conv4_x: 212 tiles, stream-K
    lib.emadapt_conv_fwd_x6p_workspace_bytes.restype = ctypes.c_size_t
    lib.emadapt_conv_dgrad_x6p_workspace_bytes.restype = ctypes.c_size_t
    rng = np.random.default_rng(1)
    kinds = set()
    for _ in range(200):
        B, H, W = int(rng.integers(1, 17)), int(rng.integers(5, 90)), int(rng.integers(5, 90))
        Cin, Cout = int(rng.choice([16, 64, 256, 512, 1024])), int(rng.choice([48, 256, 320, 512, 2048, 4096]))
        k, dil = int(rng.choice([1, 3, 4])), int(rng.choice([1, 2, 4]))
        Mx = B * H * W
        cus = int(rng.choice([0, 3, 7, 64, 256]))
        g = cus or 256
        sk_ws = -(-g * 4 // 256) * 256 + g * 256 * 256 * 4   # arrival words + one 256 KB accumulator image per workgroup
        with ops.tuned(plan_cus=cus):
            for dgrad in (False, True):
                d = ops.conv_desc(B, H, W, Cin, Cout, k, k, dil)
                ws = (lib.emadapt_conv_dgrad_x6p_workspace_bytes if dgrad else lib.emadapt_conv_fwd_x6p_workspace_bytes)(ctypes.byref(d))
                if not ops.conv_x6p_supported(d, int(dgrad)):
                    assert ws == 0
                    continue
                N = Cin if dgrad else Cout
                rows, sk, levels, classes, tail = ops.conv_x6p_plan(d, dgrad)
                table = -(-Mx // 256) * 256 * 4 if classes else 0
                if sk:
                    kinds.add("packed" if classes else "stream-k")
                    assert rows == Mx and levels != 1 and ws == (sk_ws if levels else 0) + table
                else:
                    assert classes == 0 and levels == 0 and 0 < rows <= Mx
                    if rows < Mx:
                        kinds.add("tail")
                        assert rows % 256 == 0 and tail != 1 and ws == (sk_ws if tail > 0 else 0)
                    else:
                        kinds.add("plain")
                        assert ws == 0
    assert kinds == {"plain", "stream-k", "packed", "tail"}, kinds


def test_executed_fraction_and_new_plan_queries_without_a_device():
    """Host-only arithmetic of the C ABI added in round 4: executed / algorithmic FLOPs per shape (what bench.py's frac_executed is
    built from), the bf16 plans, the LDS-DMA bf16 kernels' support rules and plans, and the batch-sliced workspace sizes of tensors
    of 2 GiB and more."""
    import ctypes
    from em_adapt_tensorflow_amd import _lib, ops
    lib = _lib.load()
    f = lambda *a, **k: ops.conv_executed_fraction(*a, **k)
    # fc6 (4x4 rate 4 on 41x41, deeplab.py:92,145): 16 taps, rows / columns -6 and +6 outside for 6 of 41 positions each
    exact = ((41 - 6) + (41 - 2) + (41 - 2) + (41 - 6)) ** 2 / (16 * 41 * 41)
    assert abs(f(16, 41, 41, 512, 4096, 4, 4, "wgrad") - exact) < 1e-12               # per-tap pixel boxes: exact
    assert exact <= f(16, 41, 41, 512, 4096, 4, 4, "fwd") < exact + 0.02               # packed rows: tiles straddling classes pay the union
    assert f(16, 41, 41, 512, 4096, 4, 4, "fwd") == f(16, 41, 41, 512, 4096, 4, 4, "dgrad")
    assert f(16, 41, 41, 4096, 4096, 1, 1, "fwd") == 1.0 and f(16, 41, 41, 4096, 4096, 1, 1, "wgrad") == 1.0   # 1x1: nothing to skip
    assert f(16, 41, 41, 512, 512, 3, 1, "fwd") == 1.0                                 # rate-1 3x3: below the 5 % packing threshold ...
    assert abs(f(16, 41, 41, 512, 512, 3, 1, "wgrad") - (40 + 41 + 40) ** 2 / (9 * 41 * 41)) < 1e-12   # ... but its filter gradient skips
    assert f(16, 41, 41, 512, 512, 3, 2, "fwd") < 0.95                                 # conv5_x, rate 2: packed
    assert f(16, 41, 41, 512, 4096, 4, 4, "fwd", "bf16") == 1.0                        # the bf16 kernels walk the dense reduction
    assert f(16, 41, 41, 512, 4096, 4, 4, "wgrad", "bf16x6") == f(16, 41, 41, 512, 4096, 4, 4, "wgrad")
    with ops.tuned(fp32_loader=2):                                                     # the dense plans (tests, A/B): nothing skipped
        assert f(16, 41, 41, 512, 4096, 4, 4, "fwd") == 1.0 and f(16, 41, 41, 512, 4096, 4, 4, "wgrad") == 1.0
    # bf16 plans at the benchmark shapes: 256 x 256 tiles for the wide layers, 128 x 64 for the 64-column ones (round 4)
    assert ops.conv_bf16_plan(16, 41, 41, 4096, 4096, 1, 1)[:2] == (256, 256)
    # (round 5: conv1_2 -- 3 x 3, 64 -> 64 channels -- on the direct kernel, 16 x 16-pixel tiles; any forced tile switches it off)
    assert ops.conv_bf16_plan(16, 321, 321, 64, 64, 3, 1) == (4, 16, 16) and ops.conv_bf16_plan(16, 321, 321, 64, 64, 3, 1, "dgrad") == (4, 16, 16)
    with ops.tuned(bf16_tile=128):
        assert ops.conv_bf16_plan(16, 321, 321, 64, 64, 3, 1)[0] == 128
    assert ops.conv_bf16_plan(16, 321, 321, 64, 64, 3, 2)[:2] == (128, 64)
    assert ops.conv_bf16_plan(16, 161, 161, 128, 64, 3, 1, "dgrad")[:2] == (128, 128)
    assert ops.conv_bf16_plan(16, 161, 161, 64, 128, 3, 1, "dgrad")[:2] == (128, 64)   # the data gradient's columns are Cin
    # LDS-DMA bf16 kernels: gathered channels % 32, stream-K where the tiles are not whole rounds
    d = ops.conv_desc(16, 41, 41, 512, 512, 3, 3, 1)
    assert ops.conv_bf16t_supported(d, "fwd") and ops.conv_bf16t_supported(d, "dgrad")
    assert not ops.conv_bf16t_supported(ops.conv_desc(16, 321, 321, 4, 64, 3, 3, 1), "fwd")            # conv1_1: 4 input channels
    rows, sk, levels, tail = ops.conv_bf16t_plan(16, 41, 41, 512, 512, 3, 1)
    assert rows == 16 * 41 * 41 and sk == 0 and levels == 0                             # 212 tiles on 256 CUs: one round of plain tiles (round 5)
    rows, sk, levels, tail = ops.conv_bf16t_plan(16, 81, 81, 256, 256, 3, 1)
    assert rows == 16 * 81 * 81 and sk == 1 and levels >= 2                             # 411 tiles: stream-K, shared tiles combined in-launch
    d3, d41 = ops.conv_desc(16, 81, 81, 256, 256, 3, 3, 1), ops.conv_desc(16, 41, 41, 256, 512, 3, 3, 1)
    assert ops.conv_bf16t_pays(d, "fwd") and ops.conv_bf16t_pays(d, "dgrad") and ops.conv_bf16t_pays(d3, "fwd") and not ops.conv_bf16t_pays(d41, "fwd")
    assert ops.conv_bf16t_pays(ops.conv_desc(16, 41, 41, 512, 4096, 4, 4, 4), "dgrad")   # fc6's data gradient: 1696 K steps per CU
    assert not ops.conv_bf16t_pays(ops.conv_desc(16, 161, 161, 128, 128, 3, 3, 1), "fwd")   # narrower than a tile
    rows, sk, levels, tail = ops.conv_bf16t_plan(16, 41, 41, 4096, 4096, 1, 1)
    assert rows == 16 * 41 * 41 and sk == 0 and levels == 0                             # 6.6 rounds: plain tiles all the way (round 5; rounds 4-5: + a stream-K tail)
    rows, sk, levels, tail = ops.conv_bf16t_plan(6, 41, 41, 4096, 4096, 1, 1)
    assert rows == 8192 and sk == 0 and tail >= 1                                      # 2.5 rounds: 2 whole rounds + a stream-K tail
    # tensors of 2 GiB and more: one launch is refused by the 32-bit offsets, the entries slice the batch; workspace = the largest slice's
    big = ops.conv_desc(129, 256, 256, 32, 64, 3, 3, 1)
    one = ops.conv_desc(127, 256, 256, 32, 64, 3, 3, 1)
    assert 129 * 256 * 256 * 64 * 4 >= 2 ** 31 > 127 * 256 * 256 * 64 * 4
    assert lib.emadapt_conv_fwd_workspace_bytes(ctypes.byref(big)) == lib.emadapt_conv_fwd_workspace_bytes(ctypes.byref(one))
    kn = 3 * 3 * 32 * 64 * 4
    assert lib.emadapt_conv_wgrad_workspace_bytes(ctypes.byref(big)) >= lib.emadapt_conv_wgrad_workspace_bytes(ctypes.byref(one)) + 2 * kn
    plan = (ctypes.c_int * 4)()
    assert lib.emadapt_conv_plan(ctypes.byref(big), 0, plan) == -2                      # (the whole-tensor plan query still says "unsupported")
