#!/bin/bash
R=$PWD; O=$R/gpurun_out/r5c; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
YOLO_BWD_OVERLAP=0 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 $R/scripts/host_enqueue.py 6 > $O/prof.log 2>&1 || echo "prof failed"
python3 - <<PY
import csv,glob,collections
rows=[]
for f in glob.glob("$O/kt/**/*kernel_trace.csv", recursive=True): rows+=list(csv.DictReader(open(f)))
for kn in ("bn_finalize_kernel","bn_bwd_sum_kernel","bn_infer_bound"):
    agg=collections.defaultdict(list)
    for r in rows:
        if kn in r["Kernel_Name"]:
            agg[int(r.get("Grid_Size_X", r.get("Grid_Size",0)))].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
    for g in sorted(agg):
        v=sorted(agg[g]); print(kn, "grid", g, "n", len(v), "median us", round(v[len(v)//2],1))
PY
rm -rf $O/kt
