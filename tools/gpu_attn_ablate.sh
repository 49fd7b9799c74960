# Kernel-trace timing of the d = 40 attention kernels with parts of their inner loops compiled out (make -C diffusion_pullback_amd/csrc ablate_attn; WRONG results, timing only):
#   gpurun -- 'bash tools/gpu_attn_ablate.sh TAG'   -> gpurun_out/TAG_attn_ablate.txt  (average us per launch of the three kernels, per build)
TAG=${1:-attabl}
R=$PWD
cd /tmp && export TMPDIR=/tmp
for v in base 1 2 3 4 7; do
  if [ $v = base ]; then L=""; else L="DPB_LIB=$R/diffusion_pullback_amd/csrc/build/attabl$v/libdpb.so"; fi
  env $L rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_$v -o t -- python $R/bench.py --steps 36 --warmup 12 --profile-run > /dev/null 2>&1
  f=$(ls $R/gpurun_out/${TAG}_$v/*/*kernel_stats.csv $R/gpurun_out/${TAG}_$v/*kernel_stats.csv 2>/dev/null | head -1)
  echo "== build $v" >> $R/gpurun_out/${TAG}_attn_ablate.txt
  python $R/tools/kernel_avgs.py $f 48 70 | grep -E "^total|attn_jvp_kernel<40|attn_adj_kv_shared|attn_adj_q_multi" >> $R/gpurun_out/${TAG}_attn_ablate.txt
  rm -rf $R/gpurun_out/${TAG}_$v
done
cd $R; cat gpurun_out/${TAG}_attn_ablate.txt
