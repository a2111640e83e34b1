"""Sum rocprofv3 --pmc counters per kernel: pmc_summary.py <dir> [kernel substring]"""
import collections, csv, glob, sys
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for f in sorted(glob.glob(f"{sys.argv[1]}/**/*counter_collection.csv", recursive=True)):
    a = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            k = (r["Kernel_Name"][:60], r["Counter_Name"])
            a[k][0] += float(r["Counter_Value"]); a[k][1] += 1
    for (kn, cn), (v, n) in sorted(a.items()):
        print(f"{kn:60s} {cn:32s} total {v:16.0f} launches {n:4d} per-launch {v/n:14.1f}")
