#!/bin/bash
R=$PWD; O=$R/gpurun_out/r4l; mkdir -p $O; export TMPDIR=/tmp
python scripts/wgrad_1x1_probe.py 2>&1 | grep -v amdgpu
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python3 $R/scripts/wgrad_1x1_probe.py 10 > $O/prof.log 2>&1 || echo "prof failed"
cp $O/ks/*/*kernel_stats.csv $O/ks.csv 2>/dev/null; cp $O/ks/*/*kernel_trace.csv $O/trace.csv 2>/dev/null; rm -rf $O/ks
python3 - <<PY
import csv,collections
rows=list(csv.DictReader(open("$O/trace.csv")))
agg=collections.OrderedDict()
for r in rows:
    n=r["Kernel_Name"]
    if "wgrad" not in n: continue
    key=(n[:60], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size",""))
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    agg.setdefault(key,[]).append(d)
for k,v in agg.items():
    v.sort(); print(k, len(v), "median us", round(v[len(v)//2],1))
PY
