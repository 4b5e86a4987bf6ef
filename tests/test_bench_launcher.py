"""`python bench.py --gpus N` started without a launcher (the driver's command) is its own launcher: bench.launch_workers
starts N fresh workers with the torch.distributed.run environment, relays rank 0's stdout and propagates failure.  Exercised
here without a GPU: two gloo ranks running a stub step (the real step has no CPU fallback), a failing rank, a hung rank, and
bench.py itself on this GPU-less box (the parent must stay GPU-free; the workers must fail fast and say why)."""
import io
import json
import os
import subprocess
import sys
import textwrap
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import bench  # noqa: E402

STUB = textwrap.dedent("""
    import json, os, sys, time
    mode = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert os.environ["LOCAL_RANK"] == str(rank) and os.environ["MASTER_ADDR"] == "127.0.0.1" and os.environ["MASTER_PORT"]
    if mode == "fail" and rank == 1:
        sys.exit(7)
    if mode == "fail5" and rank == 5:      # (the eight-worker job: a rank in the middle dies before the rendezvous)
        sys.exit(9)
    if mode in ("fail", "hang", "fail5") and rank == 0:
        open(sys.argv[2], "w").write(str(os.getpid()))
        time.sleep(600)
    import torch, torch.distributed as dist
    sys.stdout.flush(); saved = os.dup(1); os.dup2(2, 1)       # like bench.py: backend banners go to stderr
    dist.init_process_group(backend="gloo")
    dist.barrier()
    os.dup2(saved, 1); os.close(saved)
    t = torch.tensor([float(rank + 1)])
    for _ in range(3):                       # the stub step: one all-reduce per step
        dist.all_reduce(t.clone())
    dist.all_reduce(t)
    print("rank %d chatter that must not reach the job's stdout" % rank, file=sys.stdout if rank else sys.stderr)
    if rank == 0:
        print(json.dumps({"n_gpus": dist.get_world_size(), "sum": float(t.item())}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
""")


def _stub(tmp_path):
    p = tmp_path / "stub_worker.py"
    p.write_text(STUB)
    return str(p)


def test_two_gloo_ranks_relay_exactly_rank0s_line(tmp_path):
    out = io.StringIO()
    rc = bench.launch_workers(2, [sys.executable, _stub(tmp_path), "ok"], timeout=120, out=out)
    assert rc == 0
    lines = [l for l in out.getvalue().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d == {"n_gpus": 2, "sum": 3.0}


def test_eight_gloo_ranks_relay_exactly_rank0s_line(tmp_path):
    """The node the scaling bench runs on has eight GPUs: the launcher with EIGHT workers (rendezvous, environment, one relayed
    line) -- nothing here needs a GPU, so the W = 8 launcher path is not an unknown on the day SCALE is not skipped."""
    out = io.StringIO()
    rc = bench.launch_workers(8, [sys.executable, _stub(tmp_path), "ok"], timeout=300, out=out)
    assert rc == 0
    lines = [l for l in out.getvalue().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    assert json.loads(lines[0]) == {"n_gpus": 8, "sum": 36.0}


def test_rank_5_of_8_dying_ends_the_other_seven(tmp_path):
    pidfile = tmp_path / "rank0.pid"
    t0 = time.monotonic()
    rc = bench.launch_workers(8, [sys.executable, _stub(tmp_path), "fail5", str(pidfile)], timeout=300, out=io.StringIO())
    assert rc == 9
    assert time.monotonic() - t0 < 120         # did not wait for rank 0's sleep or for the others' rendezvous time-out
    assert _gone(int(pidfile.read_text()))


def _gone(pid):
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return True
    # a zombie of another parent would still answer; ours were reaped by launch_workers
    try:
        return open("/proc/%d/stat" % pid).read().split(")")[1].split()[0] == "Z"
    except OSError:
        return True


def test_failing_rank_ends_the_others_and_sets_the_exit_code(tmp_path):
    pidfile = tmp_path / "rank0.pid"
    t0 = time.monotonic()
    rc = bench.launch_workers(2, [sys.executable, _stub(tmp_path), "fail", str(pidfile)], timeout=120, out=io.StringIO())
    assert rc == 7
    assert time.monotonic() - t0 < 60          # did not wait for rank 0's sleep
    assert _gone(int(pidfile.read_text()))


def test_hung_job_is_ended_at_the_limit(tmp_path):
    pidfile = tmp_path / "rank0.pid"
    rc = bench.launch_workers(1, [sys.executable, _stub(tmp_path), "hang", str(pidfile)], timeout=3, out=io.StringIO())
    assert rc == 124
    assert _gone(int(pidfile.read_text()))


def test_parent_of_gpus_n_never_imports_torch_or_the_library():
    code = textwrap.dedent("""
        import sys
        sys.argv = ["bench.py", "--gpus", "4", "--steps", "2"]
        sys.path.insert(0, %r)
        import bench
        seen = {}
        def fake(n, cmd, **kw):
            seen.update(n=n, cmd=cmd, torch="torch" in sys.modules, lib=any("em_adapt" in m for m in sys.modules))
            return 0
        bench.launch_workers = fake
        try:
            bench.main()
        except SystemExit as e:
            assert e.code == 0, e.code
        assert seen["n"] == 4 and seen["cmd"][1].endswith("bench.py") and seen["cmd"][2:] == sys.argv[1:], seen
        assert not seen["torch"] and not seen["lib"], seen
    """ % ROOT)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert p.returncode == 0, p.stderr[-2000:]


def test_bench_gpus_2_without_two_gpus_fails_fast_with_a_reason():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("two GPUs visible: the real job would run")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--launch-timeout", "300"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=400)
    assert p.returncode != 0
    assert p.stdout.strip() == ""
    assert "needs 2 GPUs" in p.stderr


def test_sigterm_to_the_launcher_takes_the_workers_along(tmp_path):
    """The driver's own time limit ends `python bench.py --gpus N` with a signal: no worker may outlive the launcher."""
    import signal
    pidfile = tmp_path / "rank0.pid"
    code = "import sys; sys.path.insert(0, %r); import bench; sys.exit(bench.launch_workers(1, [sys.executable, %r, 'hang', %r]))" % (
        ROOT, _stub(tmp_path), str(pidfile))
    p = subprocess.Popen([sys.executable, "-c", code], stderr=subprocess.PIPE, text=True)
    t_end = time.monotonic() + 60
    while not pidfile.exists() or not pidfile.read_text():
        assert time.monotonic() < t_end and p.poll() is None
        time.sleep(0.05)
    p.send_signal(signal.SIGTERM)
    assert p.wait(timeout=60) == 128 + signal.SIGTERM
    assert _gone(int(pidfile.read_text()))
