# SQ counters of the yardstick product on the ring / 8-phase tiles (two separate --pmc passes, --kernel-trace only).
# gpurun -- 'bash tools/pmc_p8.sh TAG [tiles...]'  ->  gpurun_out/TAG_pmc_sq*/ , gpurun_out/TAG_pmc_summary.txt
TAG=${1:-p8}; shift
R=$PWD
cd /tmp && export TMPDIR=/tmp
B="python $R/tools/gpu_p8_ab.py pmc $*"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_pmc_sq -o sq -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_pmc_sq2 -o sq2 -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_trace -o t -- $B > /dev/null 2>&1
cd $R
python tools/summarize_pmc.py gpurun_out $TAG > gpurun_out/${TAG}_pmc_summary.txt 2>&1; head -20 gpurun_out/${TAG}_pmc_summary.txt
python - <<PY
import csv, glob, re
for f in glob.glob("gpurun_out/${TAG}_trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm" in r["Name"]: print(r["Name"][:90], r["Calls"], r["AverageNs"])
PY
