# A/B of the round-5 changes outside the conv kernels (same weights / loss bit for bit in every line):
#   --pool-gate-tensor  the max-pool gradients read the pooled tensor as their gate   --no-fused-tiles  the refresh pass builds the tiled images
#   --no-fused-multi    one optimizer launch per layer
run() { timeout -k 10 250 python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(sys.argv[1], d['value'], d['ms_per_step'], d['final_loss'])" "[$*]"; }
for i in 1 2; do
run --dtype bf16 --no-alt --pool-gate-tensor --no-fused-tiles --no-fused-multi
run --dtype bf16 --no-alt --no-fused-tiles --no-fused-multi
run --dtype bf16 --no-alt --no-fused-multi
run --dtype bf16 --no-alt
done
run --pool-gate-tensor --no-fused-multi
run
