#!/bin/bash
# round-3 GPU batch e: where does C4 (YOLOv4-608 bs 16) spend its step? NMS kernel times; library yardstick; per-layer table
R=$PWD; O=$R/gpurun_out/r3e; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4_ks -- python3 $R/scripts/bench_configs.py c4 > $O/c4_prof.log 2>&1 || echo "c4 prof failed"
cp $O/c4_ks/*/*kernel_stats.csv $O/c4_kernel_stats.csv 2>/dev/null; rm -rf $O/c4_ks
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/nms_ks -- python3 $R/scripts/nms_profile.py > $O/nms_prof.log 2>&1 || echo "nms prof failed"
cp $O/nms_ks/*/*kernel_stats.csv $O/nms_kernel_stats.csv 2>/dev/null; rm -rf $O/nms_ks
cd $R
python scripts/kstats_summary.py $O/c4_kernel_stats.csv 40
python scripts/kstats_summary.py $O/nms_kernel_stats.csv 14
grep -v amdgpu $O/nms_prof.log | tail -5
python scripts/gemm_ceiling.py $O/gemm_ceiling.json > $O/gemm.log 2>&1; tail -3 $O/gemm.log
python scripts/layer_table.py $O/layer_table.json > $O/layer_table.log 2>&1; head -60 $O/layer_table.log | cut -c1-230
