"""Per-kernel roofline table of one power iteration from the committed profiles (no GPU needed):
   kernel-trace statistics (time), the two PMC passes (HBM bytes), the per-launch HIP-event CSV (GEMM flops).

    python tools/roofline_report.py [profiles] [round-tag] [iterations in the stats run] > profiles/rNN_roofline_report.md
    python tools/roofline_report.py DIR TAG ITERS CFG PMC_ITERS DTYPE "title"      # any other config (tools/roofline_configs.sh): files
                                                                                   # DIR/TAG_CFG_kernel_stats.csv, _gemm_launches_hip_events.csv, _pmc.json
"""
import csv
import json
import os
import re
import sys

DIR = sys.argv[1] if len(sys.argv) > 1 else "profiles"
TAG = sys.argv[2] if len(sys.argv) > 2 else "r01"
ITERS = float(sys.argv[3]) if len(sys.argv) > 3 else 48.0        # tools/gpu_session.sh: 12 warm-up + 36 timed iterations
CFG = sys.argv[4] if len(sys.argv) > 4 else None                  # None: the headline file names of rounds 1-2
PMC_ITERS = float(sys.argv[5]) if len(sys.argv) > 5 else 12.0
DTYPE = sys.argv[6] if len(sys.argv) > 6 else "bf16"
TITLE = sys.argv[7] if len(sys.argv) > 7 else "SD-1.5 mid-block, k = 5, bf16, one power iteration"
PEAK_TF, PEAK_HBM = {"fp32": 157.3}.get(DTYPE, 2500.0), 8.0        # dense MFMA TFLOP/s of the dtype, HBM TB/s (MI355X_MICROARCH.md)
KIND = {"0": "gemm_kernel", "1": "gemm_kernel", "2": "gemm_dma_kernel", "3": "gemm_dma_kernel", "4": "gemm_ring64_kernel", "5": "conv_halo_kernel", "6": "gemm_ring64_kernel", "11": "gemm_p8_kernel", "12": "gemm_wres_kernel"}


def family(name):
    n = re.sub(r"\(.*$", "", name.replace("void ", "").replace("dpb::", ""))
    return re.split(r"<", n)[0]


def main():
    n_stats, n_pmc_traffic, n_csv, n_pmc = (f"{TAG}_sd15_mid_k5_bf16_kernel_stats.csv", f"{TAG}_pmc_traffic_sd15_mid_k5_bf16.json",
                                            f"{TAG}_sd15_gemm_launches_hip_events.csv", f"{TAG}_pmc_sd15_mid_k5_bf16.json")
    if CFG:
        n_stats, n_pmc_traffic, n_csv, n_pmc = f"{TAG}_{CFG}_kernel_stats.csv", None, f"{TAG}_{CFG}_gemm_launches_hip_events.csv", f"{TAG}_{CFG}_pmc.json"
    stats = list(csv.DictReader(open(os.path.join(DIR, n_stats))))
    fam = {}
    for r in stats:
        f = fam.setdefault(family(r["Name"]), [0.0, 0])
        f[0] += float(r["TotalDurationNs"]) / 1e6 / ITERS
        f[1] += int(r["Calls"])
    if n_pmc_traffic:
        pmc = json.load(open(os.path.join(DIR, n_pmc_traffic)))["kernels"]
    else:                                                           # per-config runs: FETCH / WRITE sit in the summarize_pmc.py JSON itself
        pmc = {k: e for k, e in json.load(open(os.path.join(DIR, n_pmc)))["kernels"].items() if "fetch_kb_per_launch" in e}
    hbm = {}                                                        # family -> bytes per iteration
    pmc_iters = PMC_ITERS
    for k, v in pmc.items():
        if " / " in k or v.get("write_kb_per_launch") is None:
            continue
        b = (2.0 * v["fetch_kb_per_launch"] + v["write_kb_per_launch"]) * 1024.0 * v["launches"] / pmc_iters
        hbm[family(k)] = hbm.get(family(k), 0.0) + b
    flops, gemm_fams = {}, set()
    for r in csv.DictReader(open(os.path.join(DIR, n_csv))):
        big, mnkz = int(r["big"]), 2.0 * float(r["M"]) * float(r["N"]) * float(r["K"]) * float(r["Z"])
        if big <= 6 or big in (11, 12):
            f = KIND.get(r["big"], "gemm_kernel")
            flops[f] = flops.get(f, 0.0) + mnkz
            gemm_fams.add(f)
            continue
        # attention brackets (round 4): M = L, N = Lk, K = d, Z = heads x (samples | tangents | cotangents); algorithmic L x L x d products:
        # forward 2, tangent 5, adjoint 7 = 3 in the query-major kernel (scores, gP, gQ) + 4 in the key-major one (scores^T, gP^T, gV, gK), cross 2
        route = int(r["gather"])
        for f, n in {7: [("attn_fwd_kernel", 2)], 8: [("attn_jvp_kernel", 5)], 10: [("attn_cross_kernel", 2)],
                     9: [("attn_adj_q_multi_kernel" if route & 1 else "attn_adj_q_kernel", 3),
                         ("attn_adj_kv_shared_kernel" if route & 2 else "attn_adj_kv_kernel", 4)]}.get(big, []):
            flops[f] = flops.get(f, 0.0) + n * mnkz
    mfma = {}                                                       # family -> (MFMA busy cycles, SIMD-cycles available) from the SQ pass, if collected
    pm = os.path.join(DIR, n_pmc)
    if os.path.exists(pm):
        for k, e in json.load(open(pm))["kernels"].items():
            raw = e.get("raw", {})
            if "GRBM_GUI_ACTIVE" in raw:
                m = mfma.setdefault(family(k), [0.0, 0.0])
                m[0] += raw.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) * e["launches"]
                m[1] += 1024.0 * raw["GRBM_GUI_ACTIVE"] / 8.0 * e["launches"]
    # torch's own kernels in the trace (at::native::...) are the ENGINE BUILD -- since round 6 the weights are padded / permuted / converted on the device
    # (tape.py), once per process, outside every timed region: reported on a line of their own, not as part of an iteration
    setup = {f: fam.pop(f) for f in list(fam) if f.startswith("at::")}
    for f in setup:
        hbm.pop(f, None)
    total = sum(v[0] for v in fam.values())
    print(f"# Roofline report, {TITLE} ({TAG})\n")
    print("Sources: `rocprofv3 --kernel-trace --stats` (time), separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes (HBM bytes = "
          "(2·FETCH + WRITE)·1024), per-launch HIP events (GEMM shapes → algorithmic flops). "
          f"Peaks: {PEAK_TF:.0f} TFLOP/s dense {DTYPE} MFMA, 8 TB/s HBM.")
    print(f"Kernel time per iteration: **{total:.2f} ms**." + (f" (Not counted: {sum(v[0] for v in setup.values()) * ITERS:.1f} ms of torch kernels per process -- the engine build's "
          "weight packing on the device, outside the timed region.)" if setup else "") + "\n")
    print("| kernel family | ms / iter | share | launches / iter | algorithmic TFLOP/s | % MFMA peak | MFMA pipe busy (PMC) | HBM GB / iter | HBM TB/s | % HBM peak |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for f, (ms, calls) in sorted(fam.items(), key=lambda x: -x[1][0]):
        if ms < 0.01:
            continue
        tf = flops.get(f, 0.0) / (ms * 1e-3) / 1e12 if f in flops else None
        gb = hbm.get(f)
        tbs = gb / (ms * 1e-3) / 1e12 if gb else None
        print(f"| `{f}` | {ms:.3f} | {100 * ms / total:.1f} % | {calls / ITERS:.1f} | " + (f"{tf:.0f} | {100 * tf / PEAK_TF:.1f} % | " if tf else "– | – | ") +
              (f"{100 * mfma[f][0] / mfma[f][1]:.1f} % | " if f in mfma and mfma[f][1] > 0 and mfma[f][0] > 0 else "– | ") +
              (f"{gb / 1e9:.2f} | {tbs:.2f} | {100 * tbs / PEAK_HBM:.0f} % |" if gb else "– | – | – |"))
    # per-shape binding roof of the product launches: time at the MFMA peak vs time to move the unique operand bytes at the HBM peak
    shapes = {}
    for r in csv.DictReader(open(os.path.join(DIR, n_csv))):
        if int(r["big"]) > 6 and int(r["big"]) not in (11, 12):
            continue
        M, N, K, Z, g = int(r["M"]), int(r["N"]), int(r["K"]), int(r["Z"]), int(r["gather"])
        kin = K // 9 if g in (1, 2, 3) and K % 9 == 0 and K > 72 else K              # 3x3 convolutions read each input pixel once, not nine times
        es = 4.0 if DTYPE == "fp32" else 2.0
        e = shapes.setdefault((KIND.get(r["big"], "gemm_kernel"), g, M, N, K, Z), [0, 0.0, 2.0 * M * N * K * Z, (M * kin + N * K + M * N) * es * Z])
        e[0] += 1; e[1] += float(r["us"])
    gemm_ms = sum(fam[f][0] for f in gemm_fams if f in fam)
    gemm_fl = sum(flops[f] for f in gemm_fams)
    att_fams = [f for f in flops if f not in gemm_fams and f in fam]
    att_ms, att_fl = sum(fam[f][0] for f in att_fams), sum(flops[f] for f in att_fams)
    hbm_total = sum(hbm.values())
    print("\n'MFMA pipe busy' = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles) from the separate SQ counter pass (issued MFMAs incl. padded "
          "tile lanes; counter passes run ~1.4x slower than un-profiled ones, so it under-reads the un-profiled utilisation).")
    print(f"\nAll GEMM / convolution kernels together: {gemm_fl / 1e12:.2f} TFLOP in {gemm_ms:.2f} ms = "
          f"{gemm_fl / (gemm_ms * 1e-3) / 1e12:.0f} TFLOP/s ({100 * gemm_fl / (gemm_ms * 1e-3) / 1e12 / PEAK_TF:.1f} % of the MFMA peak).")
    if att_ms > 0:
        print(f"\nAttention kernels together (algorithmic L×L×d products per head: forward 2, tangent 5, adjoint 7 = 3 query-major + 4 key-major, "
              f"cross-attention 2): {att_fl / 1e12:.2f} TFLOP in {att_ms:.2f} ms = {att_fl / (att_ms * 1e-3) / 1e12:.0f} TFLOP/s "
              f"({100 * att_fl / (att_ms * 1e-3) / 1e12 / PEAK_TF:.1f} % of the MFMA peak); the shared-probability kernels issue fewer MFMAs than the algorithmic count.")
    if shapes:
        print("\n### Product launches by shape: which roof binds, and how close (HIP events, bracket-corrected)\n")
        print("Roof time = max(flops / MFMA peak, unique operand bytes / 8 TB/s): the K <= 640 products of the 64x64 / 32x32 levels are **HBM-bound by their operands** "
              "(arithmetic intensity ~ 100 flop/B against a ridge of 312), the rest MFMA-bound; `frac` = roof time / measured time.\n")
        print("| kernel | gather | M x N x K | launches | us each | roof us | bound | frac of the binding roof |")
        print("|---|---|---|---|---|---|---|---|")
        tot_us = sum(e[1] for e in shapes.values()); tot_roof = 0.0
        for key, e in sorted(shapes.items(), key=lambda kv: -kv[1][1]):
            t_m, t_h = e[2] / (PEAK_TF * 1e12) * 1e6, e[3] / (PEAK_HBM * 1e12) * 1e6
            tot_roof += max(t_m, t_h) * e[0]
            if e[1] / tot_us < 0.012:
                continue
            print(f"| `{key[0]}` | {key[1]} | {key[2]} x {key[3]} x {key[4]}" + (f" (x{key[5]})" if key[5] > 1 else "") +
                  f" | {e[0]} | {e[1] / e[0]:.1f} | {max(t_m, t_h):.1f} | {'mfma' if t_m > t_h else 'hbm'} | {max(t_m, t_h) / (e[1] / e[0]):.2f} |")
        print(f"\nAll product launches: {tot_us / 1e3:.2f} ms measured against {tot_roof / 1e3:.2f} ms of binding-roof time = **{tot_roof / tot_us:.2f}** of the roofline "
              "(launches below 1.2 % of the product time are summed but not listed).")
    if hbm_total > 0:
        print(f"\nWhole iteration: {hbm_total / 1e9:.2f} GB of HBM traffic in {total:.2f} ms of kernel time = {hbm_total / (total * 1e-3) / 1e12:.2f} TB/s "
              f"({100 * hbm_total / (total * 1e-3) / 1e12 / PEAK_HBM:.0f} % of the HBM peak); {(gemm_fl + att_fl) / 1e12:.2f} TFLOP (products + attention) = "
              f"{(gemm_fl + att_fl) / (total * 1e-3) / 1e12:.0f} TFLOP/s ({100 * (gemm_fl + att_fl) / (total * 1e-3) / 1e12 / PEAK_TF:.1f} % of the MFMA peak).")


if __name__ == "__main__":
    main()
