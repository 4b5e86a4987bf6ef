run() { timeout -k 10 250 python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(sys.argv[1], d['value'], d['ms_per_step'], d['final_loss'])" "[$*]"; }
run --dtype bf16 --no-alt --pool-gate-tensor
run --dtype bf16 --no-alt
run --dtype bf16 --no-alt --pool-gate-tensor
run --dtype bf16 --no-alt
run --pool-gate-tensor
run
