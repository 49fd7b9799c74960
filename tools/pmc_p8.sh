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
python - <<PY > gpurun_out/${TAG}_yardstick.txt
import csv, glob, collections
# per (kernel, grid size) = per (tile, M): median duration from the kernel trace, MFMA busy and stall shares from the SQ pass (same launch order in both passes)
def rows(pat):
    for f in glob.glob(pat, recursive=True):
        yield from csv.DictReader(open(f))
tr = collections.defaultdict(list)
for r in rows("gpurun_out/${TAG}_trace/**/*kernel_trace.csv"):
    if "gemm_" in r["Kernel_Name"]:
        tr[(r["Kernel_Name"].split("(")[0].replace("void dpb::", ""), int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
pm = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows("gpurun_out/${TAG}_pmc_sq/**/*counter_collection.csv"):
    if "gemm_" in r["Kernel_Name"]:
        k = (r["Kernel_Name"].split("(")[0].replace("void dpb::", ""), int(r["Grid_Size"]) // int(r["Workgroup_Size"]))
        pm[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE": cnt[k] += 1
print("# 2560 x 2560 plain-row products, bf16, random operands: rocprofv3 --kernel-trace (median of the launches, first two dropped) and a separate --pmc SQ pass (tools/pmc_p8.sh)")
print("# kernel | blocks | rows M | median us | TF/s | MFMA pipe busy | issue-stalled | parked | issuing")
for k in sorted(tr, key=lambda k: (k[1], k[0])):
    d = sorted(tr[k][2:] or tr[k]); us = d[len(d) // 2] / 1e3
    tilem = 128 if "<128" in k[0] else 256
    tilen = 128 if "128, 128" in k[0] or "<128,128" in k[0] else 256
    M = k[1] // (2560 // tilen) * tilem
    c = pm.get(k, {})
    wc = c.get("SQ_WAVE_CYCLES", 0) or 1
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024.0 * c.get("GRBM_GUI_ACTIVE", 1) / 8.0) if c else 0
    print(f"{k[0]:48s} | {k[1]:5d} | {M:6d} | {us:7.1f} | {2.0 * M * 2560 * 2560 / us / 1e6:5.0f} | {100 * busy:5.1f} % | {100 * c.get('SQ_WAIT_INST_ANY', 0) / wc:5.1f} % | {100 * c.get('SQ_WAIT_ANY', 0) / wc:5.1f} % | {100 * c.get('SQ_ACTIVE_INST_ANY', 0) / wc:5.1f} %")
PY
cat gpurun_out/${TAG}_yardstick.txt
