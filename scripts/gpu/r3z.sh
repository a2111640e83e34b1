#!/bin/bash
R=$PWD; O=$R/gpurun_out/r3z; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
for V in 0 1; do
  YOLO_INFER_ONEPASS=$V timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks$V -- python3 $R/scripts/infer_bs1_graph.py > $O/prof$V.log 2>&1 || echo "prof failed"
  cp $O/ks$V/*/*kernel_stats.csv $O/bs1_onepass${V}_kernel_stats.csv 2>/dev/null; rm -rf $O/ks$V
  grep "graph replay" $O/prof$V.log
  python3 $R/scripts/kstats_summary.py $O/bs1_onepass${V}_kernel_stats.csv 14
done
