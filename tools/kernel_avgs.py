"""Print ms/iteration per kernel family from a rocprofv3 --kernel-trace --stats CSV (argv[1]); argv[2] = iterations profiled."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
it = float(sys.argv[2]) if len(sys.argv) > 2 else 24.0
tot = sum(float(r['TotalDurationNs']) for r in rows) / 1e6 / it
print(f"total {tot:.3f} ms/iter")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 16]:
    n = re.sub(r"\(.*$", "", r['Name'].replace('void ', '').replace('dpb::', ''))
    print(f"{n[:48]:48s} calls {int(r['Calls']):5d}  ms/iter {float(r['TotalDurationNs'])/1e6/it:6.3f}  avg {float(r['AverageNs'])/1e3:8.1f} us")
