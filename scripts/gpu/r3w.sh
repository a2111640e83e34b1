#!/bin/bash
R=$PWD; O=$R/gpurun_out/r3w; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
YOLO_BWD_OVERLAP=0 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4_ks -- python3 $R/scripts/bench_configs.py c4 > $O/c4_prof.log 2>&1 || echo "c4 prof failed"
cp $O/c4_ks/*/*kernel_stats.csv $O/c4_serial_kernel_stats.csv 2>/dev/null; rm -rf $O/c4_ks
cd $R
grep -h images_per_s $O/c4_prof.log
python scripts/kstats_summary.py $O/c4_serial_kernel_stats.csv 45
