# MFMA-utilisation / stall counters per kernel of one headline bench run (separate passes, --kernel-trace only: see the rocprofv3 rules in
# MI355X_MICROARCH.md).   gpurun -- 'bash tools/pmc_mfma.sh TAG'   ->  gpurun_out/TAG_pmc_sq*/  + gpurun_out/TAG_pmc_mfma.json
TAG=${1:-pmc}
R=$PWD
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 12 --warmup 0 --profile-run"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_pmc_sq -o sq -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_pmc_sq2 -o sq2 -- $B > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_pmc_fetch -o f -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_pmc_write -o w -- $B > /dev/null 2>&1
cd $R
python tools/summarize_pmc.py gpurun_out $TAG > gpurun_out/${TAG}_pmc_summary.txt 2>&1; head -40 gpurun_out/${TAG}_pmc_summary.txt
