#!/bin/bash
# kernel traces of bs-1 predict with conv_small.hip for every qualifying layer (YOLO_CONV_SMALL_GRID=4096), one per tile shape
mkdir -p gpurun_out/r6ai
export TMPDIR=/tmp
R=$PWD
for T in 11 21 22; do
  YOLO_CONV_SMALL=3 YOLO_CONV_SMALL_TILE=$T timeout -k 10 300 python -m pytest tests/test_gpu_conv.py -x -q -k "inference_unit" > gpurun_out/r6ai/test_$T.log 2>&1 || { tail -30 gpurun_out/r6ai/test_$T.log; exit 1; }
  tail -1 gpurun_out/r6ai/test_$T.log
  rm -rf /tmp/r6ai
  ( cd /tmp && YOLO_CONV_SMALL=3 YOLO_CONV_SMALL_TILE=$T YOLO_CONV_SMALL_GRID=4096 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/r6ai -- python3 $R/scripts/infer_bs1_graph.py > $R/gpurun_out/r6ai/run_$T.log 2>&1 )
  f=$(find /tmp/r6ai -name "*kernel_trace.csv" | head -1)
  python3 - "$f" <<'P' > gpurun_out/r6ai/small_by_grid_$T.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "conv_small" in n:
        by[(n[n.index("<"):n.index(">") + 1], int(r.get("Grid_Size_X") or r.get("Grid_Size")) // 512)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in sorted(by):
    v = sorted(by[k]); print(k, "launches per predict", len(v) / 255.0, "median us", v[len(v)//2], "min", v[0])
P
  echo "== tile $T"; cat gpurun_out/r6ai/small_by_grid_$T.txt; tail -1 gpurun_out/r6ai/run_$T.log
done
