#!/bin/bash
# Kernel traces of the bf16 step with a tune knob at a value and at its default, same box, one library:
#   gpurun -- bash scripts/trace_ab.sh <knob> <value> [lib.so]   -> gpurun_out/trace_ab_<knob>/{on,off}/..._kernel_stats.csv + a per-kernel diff table
knob=$1; val=${2:-1}; lib=$3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
if [ -n "$lib" ]; then export EMADAPT_NO_BUILD=1; cp "$lib" em-adapt-tensorflow_amd/libemadapt_hip.so; fi
O=gpurun_out/trace_ab_$knob
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/on -- python3 bench.py --dtype bf16 --no-alt --no-cpu-baseline --no-prof --steps 10 --warmup 3 --tune $knob=$val > $O/on.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/off -- python3 bench.py --dtype bf16 --no-alt --no-cpu-baseline --no-prof --steps 10 --warmup 3 > $O/off.log 2>&1
grep '"metric"' $O/on.log | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('on  (%s): %.1f images/s %.3f ms' % ('$knob=$val', d['value'], d['ms_per_step']))"
grep '"metric"' $O/off.log | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('off (default): %.1f images/s %.3f ms' % (d['value'], d['ms_per_step']))"
python3 - $O <<'PY'
import csv, glob, sys
def load(d):
    f = glob.glob(d + "/*/*kernel_stats.csv")[0]
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3) for r in csv.DictReader(open(f))}
on, off = load(sys.argv[1] + "/on"), load(sys.argv[1] + "/off")
steps = 13.0
rows = []
for n in set(on) | set(off):
    a, b = on.get(n, (0, 0.0)), off.get(n, (0, 0.0))
    rows.append((a[1] / steps - b[1] / steps, n, a, b))
print("per step: on - off us | calls on/off | us on | us off | kernel")
for d, n, a, b in sorted(rows, key=lambda r: -abs(r[0]))[:28]:
    print("%+9.1f | %5.1f /%5.1f | %9.1f | %9.1f | %s" % (d, a[0] / steps, b[0] / steps, a[1] / steps, b[1] / steps, n[:110]))
print("total per step: on %.1f us, off %.1f us" % (sum(v[1] for v in on.values()) / steps, sum(v[1] for v in off.values()) / steps))
PY
rm -rf $O/on/*/*kernel_trace.csv $O/off/*/*kernel_trace.csv 2>/dev/null
