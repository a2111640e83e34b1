#!/bin/bash
R=$PWD; O=$R/gpurun_out/r5j; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/scripts/bench_configs.py c4 > $O/c4.log 2>&1 || echo "prof failed"
python3 $R/scripts/step_timeline.py $O/kt 1 --json $O/timeline_c4.json > $O/timeline_c4.txt 2>&1
cp $O/kt/*/*kernel_stats.csv $O/c4_kernel_stats.csv 2>/dev/null; rm -rf $O/kt
cat $O/timeline_c4.txt; grep "config" $O/c4.log
