#!/bin/bash
# Same-box A/B of two library builds at step level (the power-capped bf16 kernels differ by 10 % from box to box, so two builds are
# only comparable inside one gpurun call):   gpurun -- bash scripts/ab_lib.sh libA.so libB.so [bench flags]
A=$1; B=$2; shift 2
cd "$GRAFT_REPO_ROOT"
export EMADAPT_NO_BUILD=1
O=gpurun_out/ab_lib.txt
: > $O
cp em-adapt-tensorflow_amd/libemadapt_hip.so /tmp/lib_keep.so
for i in 1 2; do
  for L in "$A" "$B"; do
    cp "$L" em-adapt-tensorflow_amd/libemadapt_hip.so
    extra=""; [ "$L" = "$A" ] && extra="$AB_A_FLAGS"   # (flags only the first library understands, e.g. a tune key it alone has)
    timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-alt --steps ${AB_STEPS:-12} --warmup 4 $extra "$@" 2>/dev/null | grep '"metric"' > gpurun_out/ab_line.json
    python3 - "$L" <<'PY' >> $O
import json, sys
d = json.loads(open("gpurun_out/ab_line.json").read())
print("%-28s %8.2f images/s %8.3f ms/step  %s" % (sys.argv[1], d["value"], d["ms_per_step"], d["kernel_ms_per_step"]))
PY
  done
done
cp /tmp/lib_keep.so em-adapt-tensorflow_amd/libemadapt_hip.so
cat $O
