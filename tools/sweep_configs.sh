# BASELINE configs[1], [3] (one GPU's share: k=10, edit-prompt ctx, 8 samples advanced together; and the 64-sample job on this one GPU) and
# configs[4] (down/up block sweep, k=5, fp16 as BASELINE names it, bf16 next to it) on one GPU:
#   bash tools/sweep_configs.sh > profiles/rNN_config_sweep.jsonl
Q="--no-cpu-baseline --no-roofline --no-unet-forward --no-strong-leg"
python bench.py --workload ddpm256 --dtype fp32 $Q 2>/dev/null
python bench.py --k 10 --ctx edit --samples-per-gpu 8 --steps 96 --warmup 96 $Q 2>/dev/null
python bench.py --k 10 --ctx edit --samples 64 --samples-per-gpu 8 --warmup 12 $Q 2>/dev/null
python bench.py --k 5 --samples-per-gpu 4 --steps 96 --warmup 48 $Q 2>/dev/null
for dt in fp16 bf16; do for op in down up; do for i in 0 1 2 3; do
  python bench.py --dtype $dt --op $op --block-idx $i --steps 12 --warmup 12 $Q 2>/dev/null
done; done; done
python bench.py --dtype fp16 $Q 2>/dev/null
