# Kernel-trace A/B of one environment switch on the headline region:  gpurun -- 'bash tools/gpu_trace_ab.sh TAG VAR A B'
# -> gpurun_out/TAG_<VAR>_<value>_kernel_avgs.txt (per-kernel calls, ms/iteration, average us), 36 timed + 12 warm-up iterations each
TAG=$1; VAR=$2; A=$3; B=$4
R=$PWD
cd /tmp && export TMPDIR=/tmp
for v in $A $B; do
  env $VAR=$v rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_${VAR}_${v}_stats -o t -- python $R/bench.py --steps 36 --warmup 12 --profile-run > /dev/null 2>&1
  python $R/tools/kernel_avgs.py $(ls $R/gpurun_out/${TAG}_${VAR}_${v}_stats/*/*kernel_stats.csv $R/gpurun_out/${TAG}_${VAR}_${v}_stats/*kernel_stats.csv 2>/dev/null | head -1) 48 60 > $R/gpurun_out/${TAG}_${VAR}_${v}_kernel_avgs.txt 2>&1
  rm -rf $R/gpurun_out/${TAG}_${VAR}_${v}_stats
done
cd $R
