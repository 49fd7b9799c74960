"""Turn the raw rocprofv3 output of tools/refresh_profiles.sh (under gpurun_out/) into the committed summaries under
profiles/:  the --kernel-trace --stats CSV (copied as is), the per-kernel HBM traffic JSON from the two --pmc passes
(FETCH_SIZE, WRITE_SIZE; separate runs, MI355X_MICROARCH.md HBM section), the per-launch HIP-event CSV and the bench lines.

    python tools/summarize_profiles.py [gpurun_out] [profiles] [round-tag]
"""
import csv
import glob
import json
import os
import re
import shutil
import sys

SRC = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
DST = sys.argv[2] if len(sys.argv) > 2 else "profiles"
TAG = sys.argv[3] if len(sys.argv) > 3 else "r01"


def norm(name: str) -> str:
    """'void dpb::gemm_ring64_kernel<128, 128, 2, 0, 4>(dpb::GemmArgs)' -> 'gemm_ring64_kernel<128,128,2>' (the trailing
    template arguments of the GEMM kernels are the compile-time gather mode and wave count: they share one bench label)."""
    n = re.sub(r"\(.*$", "", name.replace("void ", "").replace("dpb::", "")).replace(" ", "")
    if n.startswith("conv_halo_kernel<"):
        return "conv_halo_kernel"                          # forward gather and its adjoint share one bench label
    m = re.match(r"(gemm_kernel|gemm_dma_kernel|gemm_ring64_kernel)<(.*)>$", n)
    if m:
        keep = 3 if m.group(1) != "gemm_kernel" else 4          # tile (+ stages / K chunk); the rest is gather mode (and wave count)
        n = f"{m.group(1)}<{','.join(m.group(2).split(',')[:keep])}>"
    return n


def one(pattern):
    hits = sorted(glob.glob(os.path.join(SRC, pattern), recursive=True))
    return hits[-1] if hits else None


def pmc(dirname, counter):
    path = one(f"{dirname}/**/*counter_collection.csv")
    out = {}
    if not path:
        return out
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        e = out.setdefault(norm(r["Kernel_Name"]), [0, 0.0])
        e[0] += 1
        e[1] += float(r["Counter_Value"])
    return out


def main():
    os.makedirs(DST, exist_ok=True)
    st = one("refresh_stats/**/*kernel_stats.csv")
    if st:
        shutil.copy(st, os.path.join(DST, f"{TAG}_sd15_mid_k5_bf16_kernel_stats.csv"))
    for src, dst in (("refresh_gemm_launches.csv", f"{TAG}_sd15_gemm_launches_hip_events.csv"),
                     ("refresh_bench_sd15.json", f"{TAG}_bench_sd15_mid_k5_bf16.json"),
                     ("refresh_bench_ddpm256.json", f"{TAG}_bench_ddpm256_mid_k5_fp32.json")):
        if os.path.exists(os.path.join(SRC, src)) and os.path.getsize(os.path.join(SRC, src)) > 0:
            shutil.copy(os.path.join(SRC, src), os.path.join(DST, dst))
    f, w = pmc("refresh_pmc_fetch", "FETCH_SIZE"), pmc("refresh_pmc_write", "WRITE_SIZE")
    if f:
        kernels = {}
        for k in sorted(f, key=lambda k: -f[k][1]):
            kernels[k] = {"launches": f[k][0], "fetch_kb_per_launch": f[k][1] / f[k][0],
                          "write_kb_per_launch": (w[k][1] / w[k][0]) if k in w else None}
        # bench.py labels the two BK=32 ring tiles as one kind
        a, b = kernels.get("gemm_dma_kernel<128,128,3>"), kernels.get("gemm_dma_kernel<256,128,3>")
        if a or b:
            parts = [x for x in (a, b) if x]
            n = sum(x["launches"] for x in parts)
            kernels["gemm_dma_kernel<128,128,3> / <256,128,3>"] = {
                "launches": n, "fetch_kb_per_launch": sum(x["fetch_kb_per_launch"] * x["launches"] for x in parts) / n,
                "write_kb_per_launch": sum((x["write_kb_per_launch"] or 0) * x["launches"] for x in parts) / n}
        json.dump({"_source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- python bench.py --steps 12 "
                              "--warmup 0 --no-cpu-baseline --no-roofline  (SD-1.5 mid, k=5, bf16, 1 sample); tools/refresh_profiles.sh",
                   "_units": "KB per launch as reported by rocprofv3; on gfx950 FETCH_SIZE counts 64 B per 128 B request for wide coalesced "
                             "reads (MI355X_MICROARCH.md HBM section): hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024, WRITE_SIZE uncalibrated",
                   "kernels": kernels}, open(os.path.join(DST, f"{TAG}_pmc_traffic_sd15_mid_k5_bf16.json"), "w"), indent=1)
    print("wrote", sorted(os.listdir(DST)))


if __name__ == "__main__":
    main()
