"""Per-launch averages of rocprofv3 --pmc counters (+ kernel duration from the kernel trace of the same pass) for the
kernels whose name contains <substring>: pmc_collect.py <pass dir> <substring> -> one JSON object on stdout"""
import collections, csv, glob, json, sys
d, sub = sys.argv[1], sys.argv[2]
cnt = collections.defaultdict(lambda: [0.0, 0])
name = None
for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            name = r["Kernel_Name"]
            c = cnt[r["Counter_Name"]]
            c[0] += float(r["Counter_Value"]); c[1] += 1
dur = [0.0, 0]
for f in glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            dur[0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); dur[1] += 1
out = {"kernel": name, "launches": max([c[1] for c in cnt.values()] + [0]),
       "avg_duration_us_under_profiler": round(dur[0] / dur[1] / 1e3, 2) if dur[1] else None,
       "counters_per_launch": {k: v[0] / v[1] for k, v in sorted(cnt.items())}}
print(json.dumps(out))
