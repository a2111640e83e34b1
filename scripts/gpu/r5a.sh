#!/bin/bash
R=$PWD; O=$R/gpurun_out/r5a; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python3 $R/scripts/host_enqueue.py 20 > $O/prof.log 2>&1 || echo "prof failed"
cp $O/ks/*/*kernel_stats.csv $O/ks.csv 2>/dev/null; rm -rf $O/ks
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/ks.csv")))
for r in rows:
    n=r["Name"]
    if "at::native" in n or "rocclr" in n or "Memset" in n or "fill" in n.lower():
        print(r["Calls"], round(float(r["AverageNs"])/1e3,1), "us", n[:110])
PY
