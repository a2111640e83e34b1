#!/bin/bash
R=$PWD; O=$R/gpurun_out/r4a; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
for V in 0 1 2 3; do
  YOLO_REDUCE_DBG=$V timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks$V -- python3 $R/scripts/infer_bs1_graph.py > $O/prof$V.log 2>&1 || echo "prof failed"
  cp $O/ks$V/*/*kernel_stats.csv $O/ks_$V.csv 2>/dev/null; rm -rf $O/ks$V
  echo "dbg=$V"; grep "graph replay" $O/prof$V.log
  python3 $R/scripts/kstats_summary.py $O/ks_$V.csv 3 | grep reduce
done
