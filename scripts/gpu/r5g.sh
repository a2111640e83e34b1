#!/bin/bash
R=$PWD; O=$R/gpurun_out/r5g; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_decode_nms.py tests/test_gpu_measurement.py -x -q > $O/t1.log 2>&1; echo "nms tests rc $?"; tail -15 $O/t1.log
timeout -k 10 300 python -m pytest tests/test_gpu_fullsize.py -x -q -k "decode_nms" > $O/t2.log 2>&1; echo "fullsize nms rc $?"; tail -3 $O/t2.log
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python3 $R/scripts/nms_c5_profile.py > $O/nms_c5.log 2>&1 || echo "prof failed"
grep -v "amdgpu.ids\|rocprofv3\|Opened result\|HSA version" $O/nms_c5.log
cp $O/ks/*/*kernel_stats.csv $O/nms_c5_kernel_stats.csv 2>/dev/null; rm -rf $O/ks
python3 $R/scripts/kstats_summary.py $O/nms_c5_kernel_stats.csv 40 | grep -i "nms\|total"
cd $R; python scripts/nms_profile.py 2>&1 | grep -v amdgpu
