"""bench.py's contract on a small workload: ONE JSON line on stdout carrying the driver's keys, the roofline and cpu_baseline
objects and (for the default fp32 arithmetic on one GPU) the bf16x6 side measurement.  Run as a child process, like the driver
does."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--size", "65", "--classes", "5", "--width", "16", "--batch", "2",
                        "--steps", "3", "--warmup", "1", "--cpu-batch", "1", "--cpu-steps", "1"] + list(flags),
                       cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_default_line_has_every_contract_key():
    d = _run()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "images/s" and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "fp32" and d["data"] == "synthetic"
    assert abs(d["value"] - 2 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-3          # value = images / timed seconds
    assert isinstance(d["config"]["workload"], str) and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 157.3 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert "traffic" in r and r["launches"] > 0 and r["avg_launch_ms"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "images/s" and c["value"] > 0 and c["cores"] >= 1 and isinstance(c["sample"], str)
    assert set(c["b1"]) >= {"value", "batch", "ms_per_step", "estep_share"} and c["b1"]["batch"] == 1   # one leg per batch size
    x = d["bf16x6"]                                                                        # the fp32-grade fast path, beside the headline
    assert x["unit"] == "images/s" and x["value"] > 0 and x["roofline"]["peak"] == 419.4 and abs(x["final_loss"] - d["final_loss"]) < 1e-3
    # configs[2] / [3]'s arithmetic and configs[4]'s geometry ride in the default line too (round-5 verdict: driver-visible)
    h = d["bf16"]
    assert "error" not in h and h["dtype"] == "bf16" and h["value"] > 0 and h["roofline"]["peak"] == 2516.6 and abs(h["final_loss"] - d["final_loss"]) < 5e-2
    assert set(d["configs"]) == {"largefov_321_fp32", "ref_513_c41_b10_fp32", "largefov_513_c41_b10_fp32", "ref_321_recipe_b6_accum5_fp32"}
    for name, leg in d["configs"].items():
        assert "error" not in leg and leg["value"] > 0 and leg["unit"] == "images/s" and leg["roofline"]["peak"] == 157.3, (name, leg)
        assert abs(leg["value"] - leg["images_per_step"] * 1e3 / leg["ms_per_step"]) / leg["value"] < 1e-3
    assert d["configs"]["ref_513_c41_b10_fp32"]["workload_key"]["size"] == 513 and d["configs"]["ref_513_c41_b10_fp32"]["workload_key"]["classes"] == 41
    assert d["configs"]["ref_321_recipe_b6_accum5_fp32"]["images_per_step"] == 30
    assert "event-timed" in d["kernel_ms_per_step_basis"]


def test_other_arithmetics_and_no_alt():
    # (--x6-planes always: at this size and width the automatic plan would leave every layer to the native fp32 kernels)
    d = _run("--dtype", "bf16x6", "--no-cpu-baseline", "--x6-planes", "always", "--width", "32")
    assert d["dtype"] == "bf16x6" and d["roofline"]["peak"] == 419.4 and "bf16x6" not in d and "cpu_baseline" not in d
    d = _run("--no-alt", "--no-cpu-baseline", "--no-prof")
    assert d["dtype"] == "fp32" and "bf16x6" not in d and "bf16" not in d and "configs" not in d and "roofline" not in d


def test_gpus_2_launches_itself_and_prints_one_line():
    """The driver's command for N > 1 is `python bench.py --gpus N` with no launcher around it: bench.py starts the N ranks
    itself (fresh processes, before anything touches the GPU).  Two gloo ranks share this box's one GPU here; on a multi-GPU
    node the default backend is RCCL with one rank per GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--size", "65", "--classes", "5",
                        "--width", "16", "--batch", "2", "--steps", "3", "--warmup", "1", "--launch-timeout", "500"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 4 and d["config"]["parallelism"] == "dp2" and d["scaling"] == "weak"
    assert d["allreduce"]["ranks"] == 2 and d["allreduce"]["backend"].startswith("gloo")
    r = d["ms_per_step_ranks"]
    assert r["n"] == 2 and 0 < r["min"] <= r["max"] <= d["ms_per_step"] * 1.001
    assert abs(d["value"] - 4 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-3          # whole-job images / timed seconds
    assert "cpu_baseline" not in d                                                          # N = 1 only


def test_six_ranks_share_the_gpu_with_the_all_links_exchange_and_the_bf16_payload():
    """Rehearsal of the multi-rank job on the one GPU: `python bench.py --gpus 6 --backend gloo` (six is the most processes this
    pool lets one job put on a card; the eight-worker launcher itself runs on CPU in tests/test_bench_launcher.py, W = 8 shard
    padding in tests/test_dist_gloo.py) with configs[2] / [3]'s exchange -- bf16 compute, bf16 gradient payload, direct
    (reduce-scatter + all-gather as two all-to-all steps), staging by the HIP kernels, Momentum on the gathered payload -- one line,
    n_gpus and allreduce.ranks from the communicator, the exchange's overhead measured in the same process."""
    import torch
    # (the pool's limit counts THIS process too once an earlier test has opened the card in it: five workers then)
    W = 5 if torch.cuda.is_initialized() else 6
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(W), "--backend", "gloo", "--size", "65", "--classes", "5",
                        "--width", "16", "--batch", "2", "--steps", "3", "--warmup", "1", "--dtype", "bf16", "--grad-dtype", "bf16",
                        "--launch-timeout", "500"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == W and d["config"]["global_batch"] == 2 * W and d["config"]["parallelism"] == "dp%d" % W and d["dtype"] == "bf16"
    a = d["allreduce"]
    assert a["ranks"] == W and a["algo"] == "direct" and a["grad_dtype"] == "bf16" and a["staging"].startswith("HIP kernels")
    assert a["wire_bytes_per_rank_per_update"] == int(2 * (W - 1) / W * (a["bytes_per_update"] // 4) * 2)
    assert a["overhead_ms_per_step"] is not None and a["ms_per_step_without_exchange"] > 0
    assert d["ms_per_step_ranks"]["n"] == W and np.isfinite(d["final_loss"])


def test_gpus_2_rccl_on_one_gpu_fails_fast_with_a_reason():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible: the RCCL job would run")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--launch-timeout", "300"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=400, text=True)
    assert p.returncode != 0 and p.stdout.strip() == "" and "needs 2 GPUs, 1 visible" in p.stderr
