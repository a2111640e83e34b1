#!/bin/bash
# C4 (YOLOv4-608 bs 16) one-stream kernel stats of the end-of-round code, for the next round's planning
O=$PWD/gpurun_out/r7b; mkdir -p $O; R=$PWD
cd /tmp; export TMPDIR=/tmp
YOLO_BWD_OVERLAP=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python3 $R/scripts/bench_configs.py c4 > $O/ks.log 2>&1
cp $O/ks/*/*kernel_stats.csv $O/c4_serial_kernel_stats.csv; rm -rf $O/ks
head -14 $O/c4_serial_kernel_stats.csv | cut -c1-150
