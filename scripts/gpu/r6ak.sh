#!/bin/bash
# the launch sequence of ONE bs-1 predict (hipGraph replay), in time order, with durations and the gaps between launches
mkdir -p gpurun_out/r6ak
export TMPDIR=/tmp
R=$PWD
rm -rf /tmp/r6ak
( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/r6ak -- python3 $R/scripts/infer_bs1_graph.py > $R/gpurun_out/r6ak/run.log 2>&1 )
f=$(find /tmp/r6ak -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'P' > gpurun_out/r6ak/seq.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last complete predict: find the last stem launch but one
idx = [i for i, r in enumerate(rows) if "stem_mfma" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
prev_end = None
tot = 0.0; gaps = 0.0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    prev_end = e
    n = r["Kernel_Name"]
    n = n.replace("void yolo::", "").replace("yolo::", "")[:60]
    g = int(r.get("Grid_Size_X") or r.get("Grid_Size")); w = int(r.get("Workgroup_Size_X") or r.get("Workgroup_Size"))
    print(f"{(e - s) / 1e3:7.2f} us  gap {gap:6.2f}  wg {g // w:5d} x {w:4d}  {n}")
    tot += (e - s) / 1e3; gaps += gap
print(f"launches {b - a}  kernel time {tot:.1f} us  gaps {gaps:.1f} us  span {(int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3:.1f} us")
P
tail -1 gpurun_out/r6ak/seq.txt; tail -1 gpurun_out/r6ak/run.log
