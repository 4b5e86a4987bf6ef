run() { timeout -k 10 250 python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(sys.argv[1], d['value'], d['ms_per_step'], d['final_loss'], d['roofline'].get('all_conv_tflops'), d['roofline'].get('frac'))" "[$*]"; }
R4="--dtype bf16 --bf16t never --no-fused-update --tune wgrad_gen=1,bf16_tile=1,wgrad_bias=1"
run $R4
run --dtype bf16
run $R4
run --dtype bf16
run --dtype bf16 --preset largefov --bf16t never --no-fused-update --tune wgrad_gen=1,bf16_tile=1,wgrad_bias=1
run --dtype bf16 --preset largefov
run
run --no-fused-update
