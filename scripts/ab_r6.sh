# same-box A/B of a tune knob on the bf16 step: ab_r6.sh <knob>=<value> [bench flags]   (A = the knob set, B = default)
kv=$1; shift
run() { timeout -k 10 250 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-alt --dtype bf16 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(sys.argv[1], d['value'], d['ms_per_step'], d['final_loss'], d.get('kernel_ms_per_step'))" "[$*]"; }
for i in 1 2; do
run --tune $kv "$@"
run "$@"
done
