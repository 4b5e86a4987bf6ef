#!/usr/bin/env python3
"""The whole training step with a launch-plan knob at two values, same process, same box (boxes differ by ~1 %):
usage: ab_step.py <knob> <v1> <v2> [bench.py flags...]   e.g.  ab_step.py fp32_loader 2 0 --preset largefov"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
knob, v1, v2 = sys.argv[1], sys.argv[2], sys.argv[3]
for rnd in range(2):
    for v in (v1, v2):
        code = ("import sys, runpy; sys.path.insert(0, %r); from em_adapt_tensorflow_amd import ops; ops.tune_set(%r, %s); "
                "sys.argv = ['bench.py', '--steps', '12', '--warmup', '4', '--no-cpu-baseline', '--no-prof', '--no-alt'] + %r; "
                "runpy.run_path(%r, run_name='__main__')" % (root, knob, v, sys.argv[4:], os.path.join(root, "bench.py")))
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True).stdout
        line = [l for l in out.splitlines() if l.startswith("{")][-1]
        d = json.loads(line)
        print("%s=%s: %.1f images/s, %.3f ms/step" % (knob, v, d["value"], d["ms_per_step"]), flush=True)
