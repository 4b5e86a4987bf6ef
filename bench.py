#!/usr/bin/env python3
"""Benchmark of the EM-Adapt training hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one full update of the reference's loop body (deeplab.py:265-270) on one synthetic batch:
forward through the DeepLab conv stack, on-device E-step, hand-written backward, gradient accumulate
(accum_num = 1), gradient all-reduce across ranks (N > 1) and the Momentum update.  Per GPU the workload is
BASELINE.json configs[1]: 321x321x3 inputs, batch 16, 21 classes, fp32, the network as written in the
reference (fc6 4x4 rate 4 -> 4096; `--preset largefov` selects the 3x3 rate-12 variant).  Inputs, labels,
dropout and the E-step visiting orders are resident in HBM before the timed region starts.

For N > 1 there is one rank per GPU (RCCL); every rank keeps batch 16 (weak scaling) and the value is the whole-job
images/s with the time taken as the MAX over ranks between two barriers.  Either a launcher provides the ranks
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N
...`: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* are read from the environment), or -- when `--gpus N` arrives with no
WORLD_SIZE in the environment -- this file is its own launcher: before it touches the GPU it starts N fresh worker
processes of itself, relays rank 0's line and exits non-zero if any worker did (`launch_workers`).

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     -- the dominant kernel class (conv implicit-GEMM fwd / dgrad / wgrad) timed live with HIP events
                  on the launch stream: algorithmic FLOPs / time vs the gfx950 fp32-matrix peak (157.3 TFLOP/s);
  cpu_baseline -- the CPU oracle (PyTorch-CPU restatement of the same step + NumPy E-step, a stand-in for the
                  reference's TF-CPU path, which cannot be installed here) timed on this host's cores on a
                  bounded sample (N = 1 only).
"""
import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MATRIX_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
PEAK_BF16_MATRIX_TFLOPS = 2516.6  # dense bf16: v_mfma_f32_32x32x16_bf16, 32768 FLOP per 32 clk per SIMD x 1024 SIMDs x 2.4 GHz


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU")
    ap.add_argument("--size", type=int, default=321)
    ap.add_argument("--classes", type=int, default=21)
    ap.add_argument("--preset", default="ref", choices=["ref", "largefov"])
    ap.add_argument("--width", type=int, default=64)
    ap.add_argument("--keep-prob", type=float, default=0.5)
    ap.add_argument("--no-alt", action="store_true",
                    help="with --dtype fp32 on one GPU the line also carries a short measurement of the same step with "
                         "compute_dtype bf16x6 (fp32-grade results from the bf16 matrix pipe, object \"bf16x6\"); this skips it")
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16", "bf16x6"],
                    help="arithmetic of the conv inner products: fp32 = native fp32 MFMA (BASELINE configs[1], the "
                         "headline); bf16 = bf16 operands with fp32 accumulation (configs[2]/[3]); bf16x6 = fp32-grade "
                         "results from six bf16 MFMAs per product over the exact 3-way bf16 split of both fp32 operands")
    ap.add_argument("--x6-planes", default="auto", choices=["auto", "always"],
                    help="bf16x6: auto = the plane-image kernels where they have enough work to fill the part at this batch size (small "
                         "/ narrow workloads then run on the native fp32 kernels); always = wherever they can run (tests)")
    ap.add_argument("--tune", default="", help="launch-plan overrides for A/B runs, e.g. fp32_loader=2,plan_cus=256")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", default="16,1",
                    help="images per CPU-baseline step, one leg per value (BASELINE.md section 4: batch 16 and config 1's single image)")
    ap.add_argument("--cpu-steps", type=int, default=5, help="timed CPU-baseline steps per leg (median is reported; BASELINE.md section 4: >= 5)")
    ap.add_argument("--cpu-warmup", type=int, default=2, help="untimed CPU-baseline steps per leg (BASELINE.md section 4: 2)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = cgroup quota / affinity")
    ap.add_argument("--no-prof", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--prof-every", type=int, default=4,
                    help="bracket the kernels of every n-th timed step with HIP events (the first one always).  An event "
                         "record is a marker packet between two kernels, ~3.8 us each on this stack "
                         "(profiles/r02_launch_gaps.txt): on all ~155 launches of a 13 ms bf16 step that is 7 %% of the step, "
                         "so the per-kernel clock samples steps instead of taxing all of them; 1 = every step")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal: initialise torch.distributed and the gradient exchange even with one rank")
    ap.add_argument("--overlap-trace", action="store_true",
                    help="N > 1: add per-bucket launch offsets / join stalls of the last update's gradient exchange")
    ap.add_argument("--bucket-mb", type=int, default=0,
                    help="gradient exchange bucket size in MB; 0 (default) = the library's plan: 32 MB buckets for the ring all-reduce (the LAST "
                         "bucket only leaves when backward ends, so it is what stays exposed), two buckets for the direct exchange -- fc8 + fc7 + "
                         "fc6 (77 %% of the bytes, final a third of the way into backward) and the convolutions")
    ap.add_argument("--release-at", default="none",
                    help="N > 1: layer of the backward pass at which the gradient all-reduces may start; \"none\" (default, as in "
                         "the library) = each bucket as soon as it is complete.  E.g. conv2_2 holds everything back until the "
                         "~6.5 ms of many-small-workgroup kernels at the end of backward (dist.py); compare the two by "
                         "allreduce.exposed_ms_per_step and the step time on a multi-GPU node")
    ap.add_argument("--grad-dtype", default="fp32", choices=["fp32", "bf16"],
                    help="N > 1: payload of the gradient exchange.  bf16 (configs[2] / [3]: the ~7 ms bf16 backward pass cannot "
                         "hide 260 MB) halves the bytes on every link; contributions are rounded once, ACCUMULATED IN FP32 by the "
                         "shard's owner, the sum rounded once; master weights / accumulators / Momentum stay fp32.  Implies "
                         "--allreduce-algo direct")
    ap.add_argument("--allreduce-algo", default="auto", choices=["auto", "allreduce", "direct"],
                    help="N > 1: allreduce = one RCCL all-reduce per bucket (ring: one xGMI link's bandwidth per step); direct = "
                         "reduce-scatter + all-gather as two all-to-all steps, every rank talking to all 7 peers at once, on a side "
                         "stream (dist.py); auto = allreduce for fp32 payloads, direct for bf16")
    ap.add_argument("--no-fused-update", action="store_true",
                    help="run accumulate / Momentum / clean as the reference's three ops (deeplab.py:202-205) instead of one pass per "
                         "layer (config fused_update: what ADAPT.train switches on for accum_num 1 without a gradient exchange; same "
                         "weights bit for bit)")
    ap.add_argument("--no-fused-tiles", action="store_true",
                    help="A/B (bf16): the optimizer pass does not write the tiled bf16 filter images of the LDS-DMA layers; the next "
                         "forward's refresh pass builds them from a second read of the weights (config fused_tiles = False; same bits)")
    ap.add_argument("--no-fused-multi", action="store_true",
                    help="A/B: the fused update runs one optimizer launch per layer instead of one launch over all the layers without a "
                         "pass of their own (config fused_update_multi = False; same bits)")
    ap.add_argument("--pool-gate-tensor", action="store_true",
                    help="A/B: the max-pool gradients read the pooled tensor as their ReluGrad gate (config pool_gate_idx = False) "
                         "instead of the bit the forward pool left in its index bytes; same gradients bit for bit")
    ap.add_argument("--bf16t", default="auto", choices=["auto", "always", "never"],
                    help="--dtype bf16: which layers run forward / data gradient on the LDS-DMA kernel (csrc/igemm_b16t.h)")
    ap.add_argument("--overhead-steps", type=int, default=-1,
                    help="N > 1 / --force-dist: after the timed region, this many more steps WITHOUT the gradient exchange in the same "
                         "process (-1 = as many as --steps, 0 = skip): allreduce.overhead_ms_per_step = step time with the exchange - "
                         "without (exposed_ms only sees the compute stream's wait; contention with the exchange's kernels shows here)")
    ap.add_argument("--launch-timeout", type=float, default=0.0,
                    help="N > 1 started without a launcher: seconds after which the workers are ended (0 = no limit)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only "
                                                      "for rehearsing the data-parallel flow on a single GPU)")
    return ap.parse_args()


def workload_key(args):
    """What identifies the profiled workload (profiles/*_traffic.json carry the same dict)."""
    return {"dtype": args.dtype, "preset": args.preset, "size": args.size, "classes": args.classes, "batch": args.batch,
            "width": args.width}


def host_cores():
    """CPU cores this process may really use: the cgroup quota if there is one, else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(args):
    """Time the CPU oracle on a bounded sample of the same workload (test infrastructure used as a baseline), the way BASELINE.md
    section 4 prescribes: the step of deeplab.py:265-270 at batch 16 AND at config 1's single image, all host cores, 2 warm-up + 5 timed
    steps each, medians.  The top-level value is the batch-16 leg (the batch the metric is quoted on)."""
    legs = {}
    for b in [int(x) for x in str(args.cpu_batch).split(",") if x.strip()]:
        legs["b%d" % b] = cpu_baseline_leg(args, b)
    first = legs["b16"] if "b16" in legs else next(iter(legs.values()))
    out = dict(first)
    out["protocol"] = "BASELINE.md section 4: %d warm-up + %d timed steps per leg, median; legs: %s" % (
        args.cpu_warmup, max(args.cpu_steps, 1), ", ".join(legs))
    for k, v in legs.items():
        out[k] = {kk: v[kk] for kk in ("value", "unit", "batch", "ms_per_step", "estep_share", "timed_seconds")}
    return out


def cpu_baseline_leg(args, B):
    import numpy as np
    import torch
    from oracle import deeplab_oracle as do
    from oracle import estep_oracle as eo

    cores = args.cpu_threads or host_cores()
    torch.set_num_threads(cores)
    oracle = do.DeepLabOracle(category_num=args.classes, preset=args.preset, width=args.width, dtype=torch.float32)
    params = do.init_params(oracle.specs, seed=2, std=0.01)
    plist = {k: [w.copy(), b.copy()] for k, (w, b) in params.items()}
    vel = {k: [np.zeros_like(w), np.zeros_like(b)] for k, (w, b) in params.items()}
    images, labels = do.synthetic_batch(B, args.size, args.classes, seed=0)
    rng = np.random.default_rng(1)
    S = do.output_size(args.size, oracle.specs)
    perms = eo.draw_perms(5, args.classes, seed=3)

    t_estep = [0.0]

    def one_step():
        masks = {l: (rng.random((B, S, S, plist[l][1].shape[0])) < args.keep_prob) for l in ("fc6", "fc7")}
        r = oracle.micro_step({k: (v[0], v[1]) for k, v in plist.items()}, images, labels, perms, weight_decay=1e-5,
                              keep_prob=args.keep_prob, masks=masks)
        t_estep[0] = r.get("estep_seconds", 0.0)
        acc = {k: [g[0], g[1]] for k, g in r["grads"].items()}
        do.momentum_update(plist, vel, acc, 1e-3, 0.9)

    for _ in range(args.cpu_warmup):
        one_step()
    times, shares = [], []
    for _ in range(max(args.cpu_steps, 1)):
        t0 = time.perf_counter()
        one_step()
        times.append(time.perf_counter() - t0)
        shares.append(t_estep[0] / times[-1])
    med = sorted(times)[len(times) // 2]
    return {
        "value": round(B / med, 4),
        "unit": "images/s",
        "cores": int(torch.get_num_threads()),
        "cpu_model": cpu_model(),
        "kind": "port",
        "batch": B,
        "ms_per_step": round(med * 1e3, 1),
        "timed_seconds": round(sum(times), 1),
        "estep_share": round(sorted(shares)[len(shares) // 2], 4),
        "sample": "median of %d timed steps of %d images (after %d warm-up) of the same step: %dx%d, %d classes, preset %s, "
                  "fp32, PyTorch-CPU convs + NumPy E-step oracle (stand-in for the reference's TF-CPU path); %.1f s timed" % (
                      len(times), B, args.cpu_warmup, args.size, args.size, args.classes, args.preset, sum(times)),
    }


def _traffic_profiles(key):
    """Committed PMC summaries (profiles/r*_traffic*.json, newest first) taken on exactly this workload AND on the kernel sources that
    are running: (path relative to the repository, parsed json) pairs, plus a note when only stale ones exist."""
    here = os.path.dirname(os.path.abspath(__file__))
    from em_adapt_tensorflow_amd import _build
    lib_hash = _build.source_hash()
    found, note = [], "none: no committed counter profile of this workload"
    for tpath in sorted(glob.glob(os.path.join(here, "profiles", "r*_traffic*.json")), reverse=True):
        try:
            tj = json.load(open(tpath))
        except (OSError, ValueError):
            continue
        if tj.get("key") != key:
            continue
        if tj.get("source_hash") != lib_hash:
            # the profile was taken on other kernel sources: a number carried forward would be a stale claim
            if not found and not note.startswith("stale"):   # (newest first: name the newest stale one)
                note = "stale: %s was taken on kernel sources %s, these are %s (re-run scripts/make_profiles.sh)" % (
                    os.path.relpath(tpath, here), tj.get("source_hash"), lib_hash)
            continue
        found.append((os.path.relpath(tpath, here), tj))
    return found, note


def roofline_object(prof, prof_steps, steps, every, dtype, ms, fb_flops, batch, key):
    """The `roofline` object of a line (or of a side leg) from the library's per-class HIP-event sums."""
    conv = {k: prof[k] for k in ("igemm_fwd", "igemm_dgrad", "igemm_wgrad")}
    dom = max(conv, key=lambda k: conv[k]["ms"])
    d = conv[dom]
    tf = d["flops"] / (d["ms"] * 1e-3) / 1e12 if d["ms"] > 0 else 0.0
    tf_exec = d["flops_executed"] / (d["ms"] * 1e-3) / 1e12 if d["ms"] > 0 else 0.0
    all_flops = sum(v["flops"] for v in conv.values())
    all_exec = sum(v["flops_executed"] for v in conv.values())
    all_ms = sum(v["ms"] for v in conv.values())
    peak = {"fp32": PEAK_FP32_MATRIX_TFLOPS, "bf16": PEAK_BF16_MATRIX_TFLOPS, "bf16x6": round(PEAK_BF16_MATRIX_TFLOPS / 6, 1)}[dtype]
    r = {
        "bound": "mfma",
        "kernel": dom,
        "achieved": round(tf, 2),
        "peak": peak,
        "unit": "TFLOP/s",
        "frac": round(tf / peak, 4),
        # MFMA utilisation proper: the FLOPs the launches ISSUED (the library's count: algorithmic minus the products
        # with a dilated filter's zero padding that the kernels skip) / time / peak.  `frac` is what the caller gets
        # (SURVEY 8d's algorithmic FLOPs), `frac_executed` how busy the matrix pipe was; both never above 1 together
        "frac_executed": round(tf_exec / peak, 4),
        "achieved_executed": round(tf_exec, 2),
        "executed_over_algorithmic": round(d["flops_executed"] / d["flops"], 4) if d["flops"] > 0 else None,
        "traffic": None,   # filled below from the committed PMC passes when they cover this workload
        "algorithmic_bytes": round(d["bytes"] / max(d["launches"], 1)),   # per launch: every operand and the output once
        "traffic_over_algorithmic": None,
        "launches": d["launches"],
        "timed_steps_with_events": "%d of %d (every %d%s)" % (prof_steps, steps, every, "th" if every > 3 else ""),
        "fractions_taken_on": "the event-timed steps (HIP events bracket every kernel of those steps on the launch stream)",
        "avg_launch_ms": round(d["ms"] / max(d["launches"], 1), 4),
        "flop_per_launch": d["flops"] / max(d["launches"], 1),
        "flop_accounting": "frac / achieved / flop_per_launch: algorithmic (2*M*K*N per launch, zero-padding products "
                           "included: SURVEY 8d); frac_executed / achieved_executed: the products the kernels really "
                           "issue (packed pixel classes and per-tap pixel boxes skip a dilated filter's padding)",
        "all_conv_tflops": round(all_flops / (all_ms * 1e-3) / 1e12, 2) if all_ms > 0 else None,
        "all_conv_tflops_executed": round(all_exec / (all_ms * 1e-3) / 1e12, 2) if all_ms > 0 else None,
        "all_conv_frac": round(all_flops / (all_ms * 1e-3) / 1e12 / peak, 4) if all_ms > 0 else None,
        "all_conv_frac_executed": round(all_exec / (all_ms * 1e-3) / 1e12 / peak, 4) if all_ms > 0 else None,
        "step_tflops": round(fb_flops * batch / (ms * 1e-3) / 1e12, 2),
    }
    # HBM (L2-miss) bytes per launch of the dominant class: PMC counters need their own rocprofv3 passes
    # (scripts/make_profiles.sh); their per-class summary is committed under profiles/ and used when it was taken
    # on exactly this workload (matched on its structured key, not on prose) and on these kernel sources
    found, note = _traffic_profiles(key)
    r["traffic_source"] = note
    for rel, tj in found:
        try:
            per_step = tj["per_step"][dom]
        except KeyError:
            continue
        calls = d["launches"] / max(prof_steps, 1)
        r["traffic_source"] = rel
        r["traffic"] = round((per_step["read_bytes"] + per_step["write_bytes"]) / max(calls, 1))
        r["traffic_unit"] = "bytes per launch (HBM-side reads + writes, %s)" % rel
        if d["bytes"] > 0:
            r["traffic_over_algorithmic"] = round(r["traffic"] / (d["bytes"] / d["launches"]), 2)
        break
    return r


def estep_object(prof, key):
    e = prof["estep"]
    if e["ms"] <= 0:
        return None
    out = {"ms": round(e["ms"] / max(e["launches"], 1), 4),
           "achieved_GBps": round(e["bytes"] / (e["ms"] * 1e-3) / 1e9, 2), "bound": "hbm",
           "peak_GBps": 8000, "algorithmic_bytes": round(e["bytes"] / max(e["launches"], 1)),
           "traffic": None}
    # counter-based HBM bytes of the E-step's kernels per call (FETCH_SIZE x 2 + WRITE_SIZE: the same passes and the
    # same staleness rule as roofline.traffic)
    for rel, tj in _traffic_profiles(key)[0]:
        if "estep" in tj:
            es = tj["estep"]
            out["traffic"] = round(es["read_bytes"] + es["write_bytes"])
            out["traffic_GBps"] = round((es["read_bytes"] + es["write_bytes"]) / (e["ms"] / max(e["launches"], 1) * 1e-3) / 1e9, 2)
            out["traffic_over_algorithmic"] = round(out["traffic"] / max(out["algorithmic_bytes"], 1), 2)
            out["l2_hit_rate"] = es.get("l2_hit_rate")
            out["traffic_source"] = rel
            break
    return out


def side_leg(what, base_cfg, args, device, fence, steps, warmup, every, dtype=None, preset=None, size=None, classes=None, batch=None, accum=1):
    """One more measurement of the same step in the same process, reported BESIDE the headline (never as it): its own model, data,
    visiting orders and warm-up, `steps` timed steps bracketed like the headline's.  With accum > 1 a step is `accum` micro-batches
    and one update (the reference's own recipe: deeplab.py:288,293)."""
    import random
    import numpy as np
    import torch
    from em_adapt_tensorflow_amd import ops
    from em_adapt_tensorflow_amd.data import SyntheticData
    from em_adapt_tensorflow_amd.deeplab import ADAPT
    from em_adapt_tensorflow_amd.estep import draw_perms

    dtype, preset = dtype or args.dtype, preset or args.preset
    size, classes, batch = size or args.size, classes or args.classes, batch or args.batch
    key = {"dtype": dtype, "preset": preset, "size": size, "classes": classes, "batch": batch, "width": args.width}
    cfg = dict(base_cfg, input_size=(size, size), category_num=classes, batch_size=batch, preset=preset, compute_dtype=dtype,
               accum_num=accum, force_gradsync=False)
    leg = None
    try:
        data = SyntheticData(dict(cfg, seed=0, unique_batches=2))
        leg = ADAPT(dict(cfg, data=data))
        leg.build()
        leg.pre_train(base_lr=1e-3, weight_decay=1e-5, momentum=0.9, batch_size=batch)
        batches = data.next_batch("train", batch)
        pool = [next(batches) for _ in range(2)]
        rng = random.Random(3)
        host = np.stack([draw_perms(5, classes, rng=rng) for _ in range((warmup + steps) * accum)])
        leg._perm_pool, leg._perm_pos = torch.as_tensor(host).to(device), 0

        def step(i):
            for m in range(accum):
                x, y = pool[(i * accum + m) % len(pool)]
                leg.micro_step(x, y, keep_prob=args.keep_prob)
            leg.apply_update()

        for i in range(warmup):
            step(i)
        fence()
        ops.prof_reset()
        prof_steps = 0
        t0 = time.perf_counter()
        for i in range(steps):
            ops.prof_enable(i % every == 0)
            prof_steps += i % every == 0
            step(warmup + i)
        fence()
        dt = time.perf_counter() - t0
        ops.prof_enable(False)
        prof = ops.prof_collect()
        ops.prof_reset()
        loss = float(leg.loss["norm"].cpu())
        ms = dt / max(steps, 1) * 1e3
        _, fb_flops = leg.flops_per_image()
        r = roofline_object(prof, prof_steps, steps, every, dtype, ms / accum, fb_flops, batch, key)
        out = {
            "what": what,
            "value": round(batch * accum * steps / dt, 3), "unit": "images/s",
            "ms_per_step": round(ms, 3), "steps": steps, "warmup": warmup,
            "images_per_step": batch * accum,
            "workload_key": dict(key, accum_num=accum),
            "dtype": dtype,
            "final_loss": round(loss, 5) if np.isfinite(loss) else None,
            "gflop_per_image_fwd_bwd": round(fb_flops / 1e9, 1),
            "roofline": {k: r[k] for k in ("kernel", "achieved", "peak", "unit", "frac", "frac_executed", "all_conv_tflops", "all_conv_frac",
                                           "all_conv_frac_executed", "traffic", "algorithmic_bytes", "traffic_over_algorithmic",
                                           "traffic_source", "avg_launch_ms", "launches", "timed_steps_with_events")},
        }
        es = estep_object(prof, key)
        if es is not None:
            out["estep_ms"] = es["ms"]
        if not np.isfinite(loss):
            out["error"] = "non-finite loss"
        return out
    except Exception as e:   # a side measurement must never cost the headline line
        return {"what": what, "error": "%s: %s" % (type(e).__name__, e)}
    finally:
        del leg
        try:
            ops.prof_enable(False)
            ops.prof_reset()
            torch.cuda.empty_cache()
        except Exception:
            pass


def launch_workers(n, worker_cmd, timeout=None, env=None, out=None):
    """`python bench.py --gpus N` without a launcher: start N fresh worker processes (one rank per GPU), relay rank 0's
    stdout (the one JSON line), return 0 only if every worker did.  Runs BEFORE anything in this process touches the GPU:
    the parent imports neither torch nor the HIP library, so the children are ordinary fork+exec of a GPU-free process.
    When a worker fails the others are ended (exact PIDs, SIGTERM then SIGKILL) and its exit code is returned.

    worker_cmd -- argv of one worker; RANK / LOCAL_RANK / WORLD_SIZE / LOCAL_WORLD_SIZE / MASTER_ADDR / MASTER_PORT are
                  supplied through the environment exactly as torch.distributed.run would.
    """
    import signal
    import socket
    import subprocess
    import threading

    out = out or sys.stdout
    base = dict(os.environ if env is None else env)
    if "MASTER_PORT" not in base:
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:   # a free port of the loopback interface
            s.bind(("127.0.0.1", 0))
            base["MASTER_PORT"] = str(s.getsockname()[1])
    base["MASTER_ADDR"] = base.get("MASTER_ADDR", "127.0.0.1")
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on these hosts
    base.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // n)))
    procs = []
    try:
        for r in range(n):
            e = dict(base, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                     GROUP_RANK="0", ROLE_RANK=str(r), EMADAPT_BENCH_WORKER="1")
            # rank 0's stdout is the job's stdout (relayed line by line); the other ranks may only talk on stderr
            procs.append(subprocess.Popen(list(worker_cmd), env=e, stdout=subprocess.PIPE if r == 0 else sys.stderr,
                                          stderr=None, text=(r == 0) or None))
    except OSError:
        for p in procs:
            p.kill()
        raise

    def relay(pipe):
        for line in pipe:
            out.write(line)
            out.flush()

    t = threading.Thread(target=relay, args=(procs[0].stdout,), daemon=True)
    t.start()
    stop = []   # the launcher itself told to stop (the driver's own time limit): take the workers along
    old_handlers = {}
    if threading.current_thread() is threading.main_thread():
        for sig in (signal.SIGTERM, signal.SIGINT):
            old_handlers[sig] = signal.signal(sig, lambda signum, frame: stop.append(signum))
    deadline = None if not timeout else time.monotonic() + timeout
    rc, failed = 0, None
    live = set(range(n))
    while live and failed is None:
        for r in sorted(live):
            c = procs[r].poll()
            if c is None:
                continue
            live.discard(r)
            if c != 0:
                failed, rc = r, (c if c > 0 else 128 - c)
                break
        if deadline is not None and time.monotonic() > deadline and live:
            failed, rc = min(live), 124
        if stop and live and failed is None:
            failed, rc = min(live), 128 + stop[0]
        if live and failed is None:
            time.sleep(0.05)
    if failed is not None:
        print("bench.py: %s; ending the %d worker(s) still running" % (
            "launcher received signal %d" % stop[0] if stop else "rank %d %s" % (
                failed, "timed out" if rc == 124 else "exited with code %d" % rc), sum(procs[r].poll() is None for r in live)),
            file=sys.stderr)
        for r in live:
            if procs[r].poll() is None:
                procs[r].send_signal(signal.SIGTERM)
        t_kill = time.monotonic() + 10
        for r in live:
            try:
                procs[r].wait(timeout=max(0.1, t_kill - time.monotonic()))
            except subprocess.TimeoutExpired:
                procs[r].kill()
                procs[r].wait()
    for sig, h in old_handlers.items():
        signal.signal(sig, h)
    t.join(timeout=5)
    return rc


def main():
    args = parse()
    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and world_env is None:
        # started without a launcher (the driver's `python bench.py --gpus N`): be the launcher.  Nothing above this line,
        # and nothing in launch_workers, imports torch or loads the HIP library.
        raise SystemExit(launch_workers(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                        timeout=args.launch_timeout or None))
    import torch

    world = int(world_env or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or args.force_dist
    if args.gpus != world and rank == 0 and distributed:
        print("note: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)
    if args.backend == "nccl" and distributed and torch.cuda.device_count() < int(os.environ.get("LOCAL_WORLD_SIZE", world)):
        # fail fast and say why (device_count() does not initialise the GPU): RCCL wants one device per rank
        raise SystemExit("bench.py --gpus %d --backend nccl needs %d GPUs, %d visible (one rank per GPU; to rehearse the "
                         "data-parallel flow on fewer GPUs use --backend gloo)" % (world, world, torch.cuda.device_count()))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X GPU (the product path has no CPU fallback)")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if args.backend == "nccl" else local_rank % max(ndev, 1)
    if dev_index >= ndev:
        raise SystemExit("rank %d needs GPU %d but only %d visible" % (rank, dev_index, ndev))
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if distributed:
        import torch.distributed as dist
        # stdout carries exactly one JSON line: RCCL prints its NCCL_DEBUG=VERSION banner (set on the GPU boxes) with a
        # bare printf when the communicator is created, so file descriptor 1 points at stderr while that happens
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:   # --force-dist started without a launcher: a one-rank job
            os.environ.setdefault("MASTER_PORT", "29531")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            if args.backend == "nccl":
                dist.init_process_group(backend="nccl", device_id=device)
            else:
                dist.init_process_group(backend=args.backend)
            dist.barrier()
            torch.cuda.synchronize(device)
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    from em_adapt_tensorflow_amd import ops
    from em_adapt_tensorflow_amd.data import SyntheticData
    from em_adapt_tensorflow_amd.deeplab import ADAPT
    from em_adapt_tensorflow_amd.estep import draw_perms
    import numpy as np
    import random

    cfg = {"input_size": (args.size, args.size), "category_num": args.classes, "accum_num": 1,
           "batch_size": args.batch, "preset": args.preset, "width": args.width, "seed": 2, "device": device,
           "compute_dtype": args.dtype, "force_gradsync": bool(args.force_dist), "bucket_mb": args.bucket_mb or None,
           "allreduce_release_at": None if args.release_at == "none" else args.release_at, "grad_dtype": args.grad_dtype, "x6_planes": args.x6_planes,
           "fused_update": not args.no_fused_update, "bf16t": args.bf16t,
           "pool_gate_idx": not args.pool_gate_tensor, "fused_tiles": not args.no_fused_tiles,
           "fused_update_multi": not args.no_fused_multi}
    if args.allreduce_algo != "auto":
        cfg["allreduce_algo"] = args.allreduce_algo
    for kv in filter(None, args.tune.split(",")):
        ops.tune_set(kv.split("=")[0], int(kv.split("=")[1]))
    data = SyntheticData(dict(cfg, seed=0, unique_batches=2))   # SyntheticData mixes the rank into its seed
    adapt = ADAPT(dict(cfg, data=data))
    adapt.build()
    adapt.pre_train(base_lr=1e-3, weight_decay=1e-5, momentum=0.9, batch_size=args.batch)
    batches = data.next_batch("train", args.batch)
    pool = [next(batches) for _ in range(2)]
    # E-step visiting orders for every step, drawn like the reference does and parked in HBM
    random.seed(3 + rank)
    total_steps = args.warmup + args.steps
    n_extra = 2 * (2 + (args.steps if args.overhead_steps < 0 else args.overhead_steps))   # (the two exchange-overhead legs, when there are any)
    host = np.stack([draw_perms(5, args.classes, rng=random) for _ in range(max(total_steps, 1) + n_extra)])
    adapt._perm_pool = torch.as_tensor(host).to(device)
    adapt._perm_pos = 0

    def step(i):
        x, y = pool[i % len(pool)]
        adapt.micro_step(x, y, keep_prob=args.keep_prob)
        adapt.apply_update()

    def fence():
        torch.cuda.synchronize(device)
        t_local = time.perf_counter()   # when THIS rank's GPU was done (per-rank spread in the line; `value` uses the fenced time)
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(device)
        return t_local

    for i in range(args.warmup):
        step(i)
    fence()
    if adapt._gradsync is not None:
        adapt._gradsync.timing = True
        adapt._gradsync.stats(reset=True)
        adapt._gradsync.trace = bool(args.overlap_trace)
    every = max(args.prof_every, 1)
    prof_steps = 0
    if not args.no_prof:
        ops.prof_reset()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if not args.no_prof:
            ops.prof_enable(i % every == 0)
            prof_steps += i % every == 0
        step(args.warmup + i)
    t_local = fence()
    dt = time.perf_counter() - t0
    prof = None
    if not args.no_prof:
        ops.prof_enable(False)
        prof = ops.prof_collect()
        ops.prof_reset()
    gs_stats = adapt._gradsync.stats() if adapt._gradsync is not None else None
    rank_ms = [(t_local - t0) / max(args.steps, 1) * 1e3]
    ms_without_exchange = ms_without_exchange_unfused = None
    n_over = args.steps if args.overhead_steps < 0 else args.overhead_steps
    if adapt._gradsync is not None and n_over > 0:
        # the same steps with the exchange switched off (same process, same buffers; the ranks' weights drift apart from here on,
        # which nothing below depends on): what the exchange COSTS the step, hidden or not.  Two legs (advisor finding, round 5):
        # without a gradient exchange the step runs the FUSED update (accumulate + Momentum + clean, one pass per layer) -- the
        # single-GPU job's step, so `overhead_ms_per_step` is what data parallelism costs all told, the loss of that fusion included;
        # the second leg switches the fusion off too (accumulate, then Momentum, as separate passes), so that
        # `overhead_vs_unfused_ms_per_step` is the exchange's launches, staging passes and contention alone
        gs_keep, adapt._gradsync = adapt._gradsync, None
        fused_keep = adapt._fused
        at = total_steps
        for unfused in (False, True):
            adapt._fused = fused_keep and not unfused
            for i in range(2):
                step(at + i)
            fence()
            t1 = time.perf_counter()
            for i in range(n_over):
                step(at + 2 + i)
            fence()
            t_leg = (time.perf_counter() - t1) / n_over * 1e3
            at += 2 + n_over
            if unfused:
                ms_without_exchange_unfused = t_leg
            else:
                ms_without_exchange = t_leg
        adapt._fused = fused_keep
        adapt._gradsync = gs_keep
    if distributed:
        from em_adapt_tensorflow_amd.dist import all_reduce_max
        # the communicator's own size, not argv's: what the collectives really ran over
        world = dist.get_world_size()
        gathered = [None] * world
        dist.all_gather_object(gathered, rank_ms[0])
        rank_ms = [float(x) for x in gathered]
        dt = all_reduce_max(dt, device)
        if ms_without_exchange is not None:
            ms_without_exchange = all_reduce_max(ms_without_exchange, device)
            ms_without_exchange_unfused = all_reduce_max(ms_without_exchange_unfused, device)
        if gs_stats is not None and gs_stats["exposed_ms"] is not None:
            gs_stats["exposed_ms_max_over_ranks"] = all_reduce_max(gs_stats["exposed_ms"], device)
    loss = float(adapt.loss["norm"].cpu())
    if not np.isfinite(loss):
        raise SystemExit("non-finite loss in the benchmark step")

    if rank == 0:
        global_batch = args.batch * world
        ms = dt / max(args.steps, 1) * 1e3
        fwd_flops, fb_flops = adapt.flops_per_image()
        out = {
            "metric": "images/sec (fwd+E-step+bwd), DeepLab-LargeFOV 321x321 bs16, 1/2/4/8 MI355X",
            "value": round(global_batch * args.steps / dt, 3),
            "unit": "images/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms, 3),
            "ms_per_step_ranks": {"min": round(min(rank_ms), 3), "max": round(max(rank_ms), 3), "n": len(rank_ms)},
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {
                "workload": "DeepLab VGG-16 (%s preset: fc6 %s) %dx%d, %d classes, batch %d per GPU, %s, "
                            "random-init N(0,0.01) weights, synthetic pixel labels, keep_prob %.2f, accum_num 1" % (
                                args.preset, "4x4 rate 4 -> 4096" if args.preset == "ref" else "3x3 rate 12 -> 1024",
                                args.size, args.size, args.classes, args.batch,
                                {"fp32": "fp32", "bf16": "bf16 conv operands / fp32 accumulate, fp32 master weights, bf16 activation storage",
                                 "bf16x6": "fp32 tensors, conv products as six bf16 MFMAs over the exact 3-way bf16 split "
                                           "(fp32-grade), fp32 accumulate"}[args.dtype], args.keep_prob),
                "workload_key": workload_key(args),
                "global_batch": global_batch,
                "parallelism": "dp%d" % world,
                "fused_update": bool(adapt._fused and adapt._gradsync is None),
                "gflop_per_image_fwd_bwd": round(fb_flops / 1e9, 1),
                "params": adapt.parameter_count(),
            },
            "final_loss": round(loss, 5),
        }
        if prof is not None:
            out["roofline"] = roofline_object(prof, prof_steps, args.steps, every, args.dtype, ms, fb_flops, args.batch, workload_key(args))
            # per event-timed step: those steps carry two marker packets per launch, so they run a little slower than the un-timed
            # ones that dominate ms_per_step -- the class sums below (and every fraction in `roofline`) are taken on the EVENT-TIMED
            # steps and may add up to more than ms_per_step, which averages all steps
            out["kernel_ms_per_step"] = {k: round(v["ms"] / max(prof_steps, 1), 3) for k, v in prof.items()}
            out["kernel_ms_per_step_basis"] = "event-timed steps only (%s); ms_per_step averages all %d steps" % (
                out["roofline"]["timed_steps_with_events"], args.steps)
            es = estep_object(prof, workload_key(args))
            if es is not None:
                out["estep"] = es
        if gs_stats is not None:
            # data-parallel evidence: how much of the gradient all-reduce was NOT hidden under backward
            out["allreduce"] = {
                "backend": args.backend + (" (RCCL)" if args.backend == "nccl" else ""),
                "ranks": gs_stats["world_size"],
                "bytes_per_update": gs_stats["bytes_per_update"],
                "algo": gs_stats["algo"], "grad_dtype": gs_stats["payload"],
                "wire_bytes_per_rank_per_update": gs_stats["wire_bytes_per_update"],
                "buckets": gs_stats["buckets"],
                "bucket_mb": args.bucket_mb,
                "release_at": args.release_at,
                "exposed_ms_per_step": None if gs_stats["exposed_ms"] is None else round(gs_stats.get("exposed_ms_max_over_ranks", gs_stats["exposed_ms"]), 3),
                "backward_window_ms": None if gs_stats["backward_window_ms"] is None else round(gs_stats["backward_window_ms"], 3),
                # step time with the exchange - the same steps without it, same process (MAX over ranks each): contention included
                "overhead_ms_per_step": None if ms_without_exchange is None else round(ms - ms_without_exchange, 3),
                "ms_per_step_without_exchange": None if ms_without_exchange is None else round(ms_without_exchange, 3),
                # ... and against the same steps with the fused update switched off as well: the exchange alone
                "overhead_vs_unfused_ms_per_step": None if ms_without_exchange_unfused is None else round(ms - ms_without_exchange_unfused, 3),
                "ms_per_step_without_exchange_unfused": None if ms_without_exchange_unfused is None else round(ms_without_exchange_unfused, 3),
                "overhead_note": "overhead_ms_per_step compares with the single-GPU step (fused accumulate + Momentum pass, which an exchange "
                                 "rules out): it includes the loss of that fusion; overhead_vs_unfused_ms_per_step does not",
                "staging": "HIP kernels (emadapt_grad_pack / _owner_sum / emadapt_sgd_momentum_payload)" if gs_stats["algo"] == "direct" else "none (RCCL all-reduce in place)",
            }
            if "trace" in gs_stats:
                out["allreduce"]["trace_last_update"] = gs_stats["trace"]
        if world == 1 and args.dtype == "fp32" and not args.no_alt:
            # Side legs, reported beside the headline and never as it (each with its own model, data and warm-up; the same number of
            # timed steps; `python bench.py --dtype ... --preset ... --size ...` prints any of them as a line of its own):
            #   bf16x6  -- DESIGN 4.4: every tensor fp32, conv products as six bf16 MFMAs over the exact 3-way split (fp32-grade)
            #   bf16    -- BASELINE configs[2] / [3]'s per-GPU arithmetic at batch 16 (bf16 operands, fp32 accumulate)
            #   configs -- BASELINE configs[4]'s geometry (513 x 513, 41 classes: `ref` at batch 10, and the LargeFOV preset it names
            #              through "atrous rate=12"), the LargeFOV preset at 321 x 321, and the reference's own recipe shape
            #              (micro-batch 6 x accum_num 5, deeplab.py:288,293), all fp32
            n_leg = max(args.steps, 1)
            out["bf16x6"] = side_leg("the same step with compute_dtype bf16x6: fp32 tensors, every conv product as six bf16 MFMAs over the exact "
                                     "3-way bf16 split of both operands, fp32 accumulation (fp32-grade: same parity bars as the native path)",
                                     cfg, args, device, fence, n_leg, args.warmup, every, dtype="bf16x6")
            out["bf16"] = side_leg("the same step with compute_dtype bf16 (BASELINE configs[2] / [3] per GPU): bf16 conv operands / fp32 "
                                   "accumulate, fp32 master weights, bf16 activation storage", cfg, args, device, fence, n_leg, args.warmup,
                                   every, dtype="bf16")
            out["configs"] = {
                "largefov_321_fp32": side_leg("LargeFOV preset (fc6 3x3 rate 12 -> 1024), 321x321, 21 classes, batch 16, fp32",
                                              cfg, args, device, fence, n_leg, args.warmup, every, preset="largefov"),
                "ref_513_c41_b10_fp32": side_leg("BASELINE configs[4] geometry on the network as written: 513x513, 41 classes, batch 10, fp32",
                                                 cfg, args, device, fence, max(n_leg // 2, 3), 2, every, size=513, classes=41, batch=10),
                "largefov_513_c41_b10_fp32": side_leg("BASELINE configs[4]: 513x513, 41 classes, fc6 3x3 rate 12, batch 10, fp32",
                                                      cfg, args, device, fence, max(n_leg // 2, 3), 2, every, preset="largefov", size=513,
                                                      classes=41, batch=10),
                "ref_321_recipe_b6_accum5_fp32": side_leg("the reference's recipe shape (deeplab.py:288,293): micro-batch 6 x accum_num 5 = 30 images "
                                                          "per update, 321x321, 21 classes, fp32; a step = 5 micro-batches + one update",
                                                          cfg, args, device, fence, max(n_leg // 2, 3), 1, every, batch=6, accum=5),
            }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
