# Same-session A/B of environment-selected variants on the headline bench (box-to-box variance is +-1.5 %: only numbers from ONE session compare).
#   gpurun -- '[BENCH_ARGS="--k 10 ..."] bash tools/ab_env.sh TAG "VAR=a VAR2=b" "VAR=c" ...'   -> gpurun_out/TAG_ab.txt  (each variant run twice, interleaved)
TAG=$1; shift
echo "# bench.py ${BENCH_ARGS:---steps 60 --warmup 12}" >> gpurun_out/${TAG}_ab.txt
for rep in 1 2; do
  for v in "$@"; do
    line=$(env $v python bench.py ${BENCH_ARGS:---steps 60 --warmup 12} --no-cpu-baseline --no-roofline --no-unet-forward --no-strong-leg --repeats 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
    echo "$v : $line" | tee -a gpurun_out/${TAG}_ab.txt
  done
done
