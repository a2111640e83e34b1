"""GPU idle time inside the timed steps, from a rocprofv3 --kernel-trace csv: union of kernel intervals vs wall.
usage: trace_gaps.py <dir> [skip_fraction]"""
import csv, glob, sys
d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
n = len(rows)
rows = rows[int(n * 0.55):]          # the last steps only (warm, steady state)
t0, t1 = rows[0][0], max(r[1] for r in rows)
busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
gaps = []
for s, e, k in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, k))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
wall = t1 - t0
print(f"kernels {len(rows)}, wall {wall/1e6:.2f} ms, busy (union) {busy/1e6:.2f} ms, idle {100*(wall-busy)/wall:.2f} %")
gaps.sort(reverse=True)
print("largest gaps (us, next kernel):")
for g, k in gaps[:12]:
    print(f"  {g/1e3:8.1f}  {k[:80]}")
import collections
hist = collections.Counter(min(int(g / 1e3), 50) for g, _ in gaps)
print("gap histogram (us: count):", sorted(hist.items())[:20], "total gap ms", sum(g for g, _ in gaps) / 1e6)
