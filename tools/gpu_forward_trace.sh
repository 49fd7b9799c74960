# Kernel-trace statistics of the DDIM / guidance loop's U-Net forward (dpb_forward) at several batches:  gpurun -- 'bash tools/gpu_forward_trace.sh TAG 2 20'
TAG=$1; shift
R=$PWD
cd /tmp && export TMPDIR=/tmp
for B in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_fwd_b${B}_stats -o k -- python $R/tools/gpu_unet_bench.py bf16 $B 20 > $R/gpurun_out/${TAG}_fwd_b${B}.txt 2>&1
  cp $(ls $R/gpurun_out/${TAG}_fwd_b${B}_stats/*/*kernel_stats.csv $R/gpurun_out/${TAG}_fwd_b${B}_stats/*kernel_stats.csv 2>/dev/null | head -1) $R/gpurun_out/${TAG}_unet_forward_b${B}_kernel_stats.csv
  rm -rf $R/gpurun_out/${TAG}_fwd_b${B}_stats
  grep forward $R/gpurun_out/${TAG}_fwd_b${B}.txt
done
cd $R
