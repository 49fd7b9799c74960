# BASELINE configs[3] (k=10, 8 samples per GPU advanced together) and configs[4] (down/up block sweep, k=5) on one GPU:
#   bash tools/sweep_configs.sh > profiles/rNN_config4_config5_sweep.jsonl
python bench.py --k 10 --samples-per-gpu 8 --steps 96 --warmup 96 --no-cpu-baseline --no-roofline 2>/dev/null
for op in down up; do for i in 0 1 2 3; do
  python bench.py --op $op --block-idx $i --steps 12 --warmup 12 --no-cpu-baseline --no-roofline 2>/dev/null
done; done
