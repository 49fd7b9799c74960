"""Per-kernel PMC summary of tools/pmc_mfma.sh:  python tools/summarize_pmc.py gpurun_out TAG  -> gpurun_out/TAG_pmc_mfma.json + a table.

MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE / 8): the share of SIMD-cycles with the matrix pipe
busy while the kernel ran.  SQ_VALU_MFMA_BUSY_CYCLES counts cycles, 32 per v_mfma_f32_32x32x16 (checked: = 32 x SQ_INSTS_MFMA in every
row); rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs (8 x the kernel's duration in cycles, checked against the kernel trace),
hence the / 8.  The SQ_WAIT_* / SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* counters count quad-cycles and are reported as shares of SQ_WAVE_CYCLES.
Counter passes run ~1.4x slower than un-profiled ones (clock + collection), so the utilisation is a lower bound of the un-profiled value.
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE correction of the guide)."""
import csv
import glob
import json
import os
import re
import sys

SRC = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
TAG = sys.argv[2] if len(sys.argv) > 2 else "pmc"


def norm(name):
    return re.sub(r"\(.*$", "", name.replace("void ", "").replace("dpb::", "")).replace(" ", "")


def load(sub):
    out = {}
    for path in glob.glob(os.path.join(SRC, f"{TAG}_{sub}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            k = norm(r["Kernel_Name"])
            d = out.setdefault(k, {})
            e = d.setdefault(r["Counter_Name"], [0, 0.0])
            e[0] += 1
            e[1] += float(r["Counter_Value"])
    return out


def main():
    sq, sq2, fe, wr = load("pmc_sq"), load("pmc_sq2"), load("pmc_fetch"), load("pmc_write")
    res = {}
    for k, c in sq.items():
        g = lambda n, src=c: src.get(n, [0, 0.0])[1]
        launches = c.get("GRBM_GUI_ACTIVE", [0, 0])[0]
        act = g("GRBM_GUI_ACTIVE")
        wc = g("SQ_WAVE_CYCLES") or 1.0
        e = {"launches": launches, "gui_active_cycles_per_launch": act / max(launches, 1),
             "mfma_util": g("SQ_VALU_MFMA_BUSY_CYCLES") / (1024.0 * act / 8.0) if act else None,
             "wait_inst_any_share": g("SQ_WAIT_INST_ANY") / wc, "wait_any_share": g("SQ_WAIT_ANY") / wc, "active_inst_any_share": g("SQ_ACTIVE_INST_ANY") / wc,
             "raw": {n: v[1] / max(v[0], 1) for n, v in c.items()}}
        if k in sq2:
            e["raw"].update({n: v[1] / max(v[0], 1) for n, v in sq2[k].items()})
        if k in fe and "FETCH_SIZE" in fe[k]:
            f = fe[k]["FETCH_SIZE"]; e["fetch_kb_per_launch"] = f[1] / max(f[0], 1)
        if k in wr and "WRITE_SIZE" in wr[k]:
            w = wr[k]["WRITE_SIZE"]; e["write_kb_per_launch"] = w[1] / max(w[0], 1)
        if "fetch_kb_per_launch" in e and "write_kb_per_launch" in e:
            e["hbm_bytes_per_launch"] = (2 * e["fetch_kb_per_launch"] + e["write_kb_per_launch"]) * 1024
        res[k] = e
    order = sorted(res, key=lambda k: -res[k]["gui_active_cycles_per_launch"] * res[k]["launches"])
    src_hash = ""
    try:                                                            # the library these counters were collected on (bench.py checks it before quoting them)
        src_hash = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "diffusion_pullback_amd", "libdpb.so.srchash")).read().strip()
    except OSError:
        pass
    json.dump({"_source": "tools/pmc_mfma.sh (rocprofv3 --pmc ... --kernel-trace, separate passes) on bench.py --steps 12 --warmup 0", "_src_hash": src_hash,
               "kernels": {k: res[k] for k in order}},
              open(os.path.join(SRC, f"{TAG}_pmc_mfma.json"), "w"), indent=1)
    print(f"{'kernel':64s} {'launches':>8s} {'Mcyc tot':>9s} {'MFMA util':>9s} {'wait_inst':>9s} {'wait_any':>9s} {'HBM MB/launch':>13s}")
    for k in order[:40]:
        e = res[k]
        mu = "-" if e["mfma_util"] is None else f"{100 * e['mfma_util']:.1f}%"
        hb = f"{e['hbm_bytes_per_launch'] / 1e6:.1f}" if "hbm_bytes_per_launch" in e else "-"
        print(f"{k[:64]:64s} {e['launches']:8d} {e['gui_active_cycles_per_launch'] * e['launches'] / 1e6:9.2f} {mu:>9s} {100 * e['wait_inst_any_share']:8.1f}% {100 * e['wait_any_share']:8.1f}% {hb:>13s}")


if __name__ == "__main__":
    main()
