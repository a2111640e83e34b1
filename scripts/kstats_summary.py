"""per-kernel totals of a rocprofv3 --kernel-trace --stats csv: name (shortened), calls, total ms, avg us, % -- top N"""
import csv, sys
path, top = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 30
rows = list(csv.DictReader(open(path)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
print(f"total kernel time {tot / 1e6:.2f} ms over {sum(int(r['Calls']) for r in rows)} launches")
for r in rows[:top]:
    n = r["Name"].replace("void yolo::", "").replace("yolo::", "")
    print(f"{float(r['TotalDurationNs']) / 1e6:9.3f} ms {int(r['Calls']):6d} x {float(r['AverageNs']) / 1e3:8.1f} us {100 * float(r['TotalDurationNs']) / tot:5.1f}%  {n[:110]}")
