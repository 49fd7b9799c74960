"""Not a test: in-pipeline kernel selection experiments.  Runs bench.py once per override set (DPB_GEMM_OVERRIDE, csrc/gemm.hip) with the
per-launch HIP-event CSV and prints, per overridden shape, the time inside the real pass (real epilogue flags, real cache state) against the
heuristic's choice, plus the end-to-end ms per iteration of every set (same session, so comparable).
    python tools/gpu_gemm_override.py [--args "--k 5 ..."] "320x1280x1280:0=65/1,1280x640x1280:0=65/1" "320x1280x1280:0=515/4" ...
"""
import collections, csv, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(override, args, tag):
    env = dict(os.environ)
    if override:
        env["DPB_GEMM_OVERRIDE"] = override
    out = []
    for rep in range(2):
        path = os.path.join(ROOT, "gpurun_out", f"ovr_{tag}_{rep}.csv")
        env["DPB_PROFILE_CSV"] = path
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-unet-forward", "--no-strong-leg", "--repeats", "3", "--steps", "36", "--warmup", "12"] + args,
                           env=env, capture_output=True, text=True)
        line = json.loads(r.stdout.strip().splitlines()[-1])
        agg = collections.OrderedDict()
        for row in csv.DictReader(open(path)):
            key = (int(row["M"]), int(row["N"]), int(row["K"]), int(row["gather"]))
            a = agg.setdefault(key, [0, 0.0, row["big"]])
            a[0] += 1
            a[1] += float(row["us"])
        out.append((line["ms_per_step"], agg))
    return out


def main():
    argv = sys.argv[1:]
    args = []
    if argv and argv[0] == "--args":
        args = argv[1].split()
        argv = argv[2:]
    base = run("", args, "base")
    print("heuristic: ms/step", [round(b[0], 3) for b in base])
    for i, ov in enumerate(argv):
        got = run(ov, args, f"s{i}")
        print(f"set {i}: {ov}\n   ms/step", [round(g[0], 3) for g in got])
        for item in ov.split(","):
            shp, rhs = item.split("=")
            mnk, g = shp.split(":")
            key = tuple(int(v) for v in mnk.split("x")) + (int(g),)
            b = [r[1].get(key) for r in base]
            n = [r[1].get(key) for r in got]
            if b[0] is None or n[0] is None:
                print(f"   {item}: shape not on the path")
                continue
            print(f"   {item:34s} n={b[0][0]:2d}  heuristic(kind {b[0][2]}) {min(x[1] for x in b) / b[0][0]:7.1f} us   override(kind {n[0][2]}) {min(x[1] for x in n) / n[0][0]:7.1f} us")


if __name__ == "__main__":
    main()
