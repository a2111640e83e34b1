"""Mean counter values per launch for kernels whose name contains argv[2], from a rocprofv3 --pmc csv dir."""
import csv, glob, sys, collections
d, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k, " ".join(f"{c}={sum(v)/len(v):.4g}(n={len(v)})" for c, v in sorted(cs.items())))
import statistics
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    ds = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"]]
    if ds:
        print("   kernel duration us: median %.1f min %.1f (n=%d)" % (statistics.median(ds) / 1e3, min(ds) / 1e3, len(ds)))
