"""Top kernels of a rocprofv3 --kernel-trace --stats csv directory. usage: kstats_top.py <dir> [n] [steps]"""
import csv, glob, sys
d = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 25; steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:n]:
    print(f"{r['Name'][:100]:100s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/steps/1e6:8.3f} ms/step {float(r['AverageNs'])/1e3:8.1f} us {100*float(r['TotalDurationNs'])/tot:5.1f}%")
print("total ms/step", tot / steps / 1e6)
