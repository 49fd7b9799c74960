"""Copy the evidence of one tools/gpu_final.sh run from gpurun_out/ (scratch) into profiles/ (tracked) under the names the tools expect:

    python tools/collect_profiles.py r02e r02        # gpurun_out/r02e_*  ->  profiles/r02_*

  r02_sd15_mid_k5_bf16_kernel_stats.csv      rocprofv3 --kernel-trace --stats of  bench.py --steps 36 --warmup 12 --no-cpu-baseline --no-roofline
  r02_sd15_gemm_launches_hip_events.csv      per-launch HIP-event durations of the GEMM kernels (bench.py roofline leg, DPB_PROFILE_CSV)
  r02_bench_sd15_mid_k5_bf16.json            the headline bench line of the same session
  r02_pmc_sd15_mid_k5_bf16.json              per-kernel PMC summary (MFMA busy, stall shares, FETCH_SIZE / WRITE_SIZE), tools/summarize_pmc.py
  r02_pmc_traffic_sd15_mid_k5_bf16.json      HBM bytes per launch per bench.py kernel label (what bench.py's `roofline.traffic` reads)
  r02_config_sweep.jsonl                     BASELINE configs[1], [3], [4] bench lines (tools/sweep_configs.sh)
"""
import json
import os
import re
import shutil
import sys

SRC_TAG = sys.argv[1]
DST_TAG = sys.argv[2] if len(sys.argv) > 2 else SRC_TAG
SRC, DST = "gpurun_out", "profiles"
TRACE_ITERS = 48.0      # tools/gpu_session.sh traces `bench.py --steps 36 --warmup 12 --profile-run`: 48 power iterations (+ 4 primal passes, ~2 % of the product time)
PMC_ITERS = 12.0        # tools/pmc_mfma.sh profiles `bench.py --steps 12 --warmup 0 --profile-run`: one sample, 12 power iterations


def label(name):
    """'gemm_ring64_kernel<128,128,2,0,4,0,0>' -> bench.py's label 'gemm_ring64_kernel<128,128,2>'"""
    if name.startswith("conv_halo_kernel<"):
        return "conv_halo_kernel"
    m = re.match(r"(gemm_kernel|gemm_dma_kernel|gemm_ring64_kernel)<(.*)>$", name)
    if m:
        keep = 3 if m.group(1) != "gemm_kernel" else 4
        return f"{m.group(1)}<{','.join(m.group(2).split(',')[:keep])}>"
    return name


def kind_of(name):
    """kernel-trace name -> the GEMM kernel kind of dpb_engine_profile_read (include/dpb.h), or None"""
    name = re.sub(r"^void\s+", "", name).replace("dpb::", "").replace(" ", "")
    m = re.match(r"(gemm_kernel|gemm_dma_kernel|gemm_ring64_kernel|conv_halo_kernel|gemm_p8_kernel|gemm_wres_kernel)<(.*)>", name)
    if not m:
        return None
    fam, par = m.group(1), m.group(2).split(",")
    if fam == "gemm_kernel":
        return 1 if par[1] == "128" else 0
    if fam == "gemm_dma_kernel":
        return 3 if par[0] == "64" else 2
    if fam == "gemm_ring64_kernel":
        return 6 if par[0] == "256" and par[1] == "256" else 4
    return 5 if fam == "conv_halo_kernel" else 12 if fam == "gemm_wres_kernel" else 11


def trace_by_kind(path):
    """per-iteration kernel time and launches of every GEMM kernel kind from a rocprofv3 --kernel-trace --stats CSV"""
    import csv
    out = {}
    for r in csv.DictReader(open(path)):
        k = kind_of(r["Name"])
        if k is None:
            continue
        e = out.setdefault(str(k), {"ms_per_iter": 0.0, "launches_per_iter": 0.0})
        e["ms_per_iter"] += float(r["TotalDurationNs"]) / 1e6 / TRACE_ITERS
        e["launches_per_iter"] += float(r["Calls"]) / TRACE_ITERS
    return out


def main():
    os.makedirs(DST, exist_ok=True)
    cp = {f"{SRC_TAG}_stats/sd15_kernel_stats.csv": f"{DST_TAG}_sd15_mid_k5_bf16_kernel_stats.csv",
          f"{SRC_TAG}_gemm_launches.csv": f"{DST_TAG}_sd15_gemm_launches_hip_events.csv",
          f"{SRC_TAG}_bench.json": f"{DST_TAG}_bench_sd15_mid_k5_bf16.json",
          f"{SRC_TAG}_pmc_mfma.json": f"{DST_TAG}_pmc_sd15_mid_k5_bf16.json",
          f"{SRC_TAG}_pmc_summary.txt": f"{DST_TAG}_pmc_summary.txt",
          f"{SRC_TAG}_config_sweep.jsonl": f"{DST_TAG}_config_sweep.jsonl",
          f"{SRC_TAG}_pytest.log": None}
    for s, d in cp.items():
        p = os.path.join(SRC, s)
        if d and os.path.exists(p) and os.path.getsize(p) > 0:
            shutil.copy(p, os.path.join(DST, d))
    pj = os.path.join(SRC, f"{SRC_TAG}_pmc_mfma.json")
    if os.path.exists(pj):
        agg = {}
        src_hash = json.load(open(pj)).get("_src_hash", "")
        for k, e in json.load(open(pj))["kernels"].items():
            if "fetch_kb_per_launch" not in e:
                continue
            a = agg.setdefault(label(k), [0, 0.0, 0.0])
            a[0] += e["launches"]; a[1] += e["fetch_kb_per_launch"] * e["launches"]; a[2] += e.get("write_kb_per_launch", 0.0) * e["launches"]
        kernels = {k: {"launches": n, "fetch_kb_per_launch": f / n, "write_kb_per_launch": w / n} for k, (n, f, w) in agg.items() if n}
        a, b = kernels.get("gemm_dma_kernel<128,128,3>"), kernels.get("gemm_dma_kernel<256,128,3>")
        if a or b:
            parts = [x for x in (a, b) if x]
            n = sum(x["launches"] for x in parts)
            kernels["gemm_dma_kernel<128,128,3> / <256,128,3>"] = {"launches": n, "fetch_kb_per_launch": sum(x["fetch_kb_per_launch"] * x["launches"] for x in parts) / n,
                                                                   "write_kb_per_launch": sum(x["write_kb_per_launch"] * x["launches"] for x in parts) / n}
        whole = sum((2.0 * e["fetch_kb_per_launch"] + e.get("write_kb_per_launch", 0.0)) * 1024.0 * e["launches"]
                    for kn, e in json.load(open(pj))["kernels"].items()
                    if "fetch_kb_per_launch" in e and not kn.startswith("at::")) / PMC_ITERS     # (torch's kernels = the engine build's weight packing, once per process)
        json.dump({"_whole_step_hbm_bytes": whole,   # every kernel of the PMC run, per power iteration (bench.py: roofline.whole_step_hbm_frac)
                   "_source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- python bench.py --steps 12 --warmup 0 "
                              "--no-cpu-baseline --no-roofline (SD-1.5 mid, k=5, bf16, 1 sample); tools/pmc_mfma.sh",
                   "_units": "KB per launch as reported by rocprofv3; gfx950: hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (MI355X_MICROARCH.md HBM section)",
                   "_trace": ({"source": f"profiles/{DST_TAG}_sd15_mid_k5_bf16_kernel_stats.csv (rocprofv3 --kernel-trace --stats of bench.py --steps 36 --warmup 12 --profile-run, "
                                         "same session and library as the counter passes): kernel time per power iteration by GEMM kernel kind "
                                         "(dpb_engine_profile_read's kinds); includes the 4 primal passes of the run (~2 % of the product time)",
                               "iterations": TRACE_ITERS, "by_kind": trace_by_kind(os.path.join(SRC, f"{SRC_TAG}_stats/sd15_kernel_stats.csv"))}
                              if os.path.exists(os.path.join(SRC, f"{SRC_TAG}_stats/sd15_kernel_stats.csv")) else None),
                   "_src_hash": src_hash,      # source hash of the libdpb.so the counters were collected on (bench.py quotes them only for that build)
                   "kernels": kernels}, open(os.path.join(DST, f"{DST_TAG}_pmc_traffic_sd15_mid_k5_bf16.json"), "w"), indent=1)
    print("profiles/:", sorted(f for f in os.listdir(DST) if f.startswith(DST_TAG)))


if __name__ == "__main__":
    main()
