#!/bin/bash
R=$PWD; O=$R/gpurun_out/r5f; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python3 $R/scripts/nms_c5_profile.py > $O/nms_c5.log 2>&1 || echo "prof failed"
cat $O/nms_c5.log | grep -v amdgpu.ids
cp $O/ks/*/*kernel_stats.csv $O/nms_c5_kernel_stats.csv 2>/dev/null; rm -rf $O/ks
python3 $R/scripts/kstats_summary.py $O/nms_c5_kernel_stats.csv 40 | grep -i "nms\|decode\|total" 
