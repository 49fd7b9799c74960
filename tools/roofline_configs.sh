# Per-config roofline evidence (BASELINE configs[1], [3], [4]): for each config the bench line + per-launch GEMM CSV, rocprofv3 kernel-trace stats and
# three separate --pmc passes (SQ MFMA-busy, FETCH_SIZE, WRITE_SIZE), summarised into gpurun_out/TAG_roofline_CFG.md by tools/roofline_report.py.
#   gpurun -- 'bash tools/roofline_configs.sh r03 [cfg ...]'      then copy gpurun_out/r03_roofline_*.md, *_kernel_stats.csv, *_pmc.json to profiles/
TAG=${1:-rXX}; shift
R=$PWD
CFGS=${@:-ddpm256_mid_k5_fp32 sd15_mid_k10x8_bf16 sd15_down0_k5_fp16 sd15_up3_k5_fp16}
for CFG in $CFGS; do
  case $CFG in
    ddpm256_mid_k5_fp32) ARGS="--workload ddpm256 --dtype fp32"; ST=12; WU=12; PST=12; SPG=1; DT=fp32; TITLE="BASELINE configs[1]: DDPM-256 mid-block, k = 5, fp32, one power iteration";;
    sd15_mid_k10x8_bf16) ARGS="--k 10 --ctx edit --samples-per-gpu 8"; ST=96; WU=96; PST=96; SPG=8; DT=bf16; TITLE="BASELINE configs[3]: SD-1.5 mid-block, k = 10, edit ctx, 8 samples advanced together, bf16, one pass over the 8 samples";;
    sd15_down0_k5_fp16) ARGS="--dtype fp16 --op down --block-idx 0"; ST=12; WU=12; PST=12; SPG=1; DT=fp16; TITLE="BASELINE configs[4]: SD-1.5 down_block_0 tap, k = 5, fp16, one power iteration";;
    sd15_up3_k5_fp16) ARGS="--dtype fp16 --op up --block-idx 3"; ST=12; WU=12; PST=12; SPG=1; DT=fp16; TITLE="BASELINE configs[4]: SD-1.5 up_block_3 tap, k = 5, fp16, one power iteration";;
    sd15_mid_k5_bf16) ARGS=""; ST=36; WU=12; PST=12; SPG=1; DT=bf16; TITLE="BASELINE configs[2]: SD-1.5 mid-block, k = 5, bf16, one power iteration";;
    *) echo "unknown config $CFG"; continue;;
  esac
  P=gpurun_out/${TAG}_${CFG}
  cd $R
  DPB_PROFILE_CSV=${P}_gemm_launches_hip_events.csv python bench.py $ARGS --steps $ST --warmup $WU --no-cpu-baseline --no-unet-forward --no-strong-leg --repeats 3 > ${P}_bench.json 2> ${P}_bench.err
  cd /tmp && export TMPDIR=/tmp
  B="python $R/bench.py $ARGS --profile-run"
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/${P}_stats -o k -- $B --steps $ST --warmup $WU > /dev/null 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/${P}_pmc_sq -o sq -- $B --steps $PST --warmup 0 > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/${P}_pmc_fetch -o f -- $B --steps $PST --warmup 0 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/${P}_pmc_write -o w -- $B --steps $PST --warmup 0 > /dev/null 2>&1
  cd $R
  cp $(ls ${P}_stats/*/*kernel_stats.csv ${P}_stats/*kernel_stats.csv 2>/dev/null | head -1) ${P}_kernel_stats.csv
  python tools/summarize_pmc.py gpurun_out ${TAG}_${CFG} > ${P}_pmc_summary.txt 2>&1
  cp ${P}_pmc_mfma.json ${P}_pmc.json
  python tools/roofline_report.py gpurun_out $TAG $(( (ST + WU) / SPG )) $CFG $(( PST / SPG )) $DT "$TITLE" > gpurun_out/${TAG}_roofline_${CFG}.md 2> ${P}_report.err
  head -12 gpurun_out/${TAG}_roofline_${CFG}.md
  rm -rf ${P}_stats ${P}_pmc_sq ${P}_pmc_fetch ${P}_pmc_write      # raw traces are large; the summaries above are what is kept
done
