#!/bin/bash
# per-kernel durations of the x-window filter gradient and its reduce (rocprofv3 kernel trace of the standalone harness)
O=$PWD/gpurun_out/r6c; mkdir -p $O; R=$PWD
cd /tmp; export TMPDIR=/tmp
export CONV_BENCH_WGRAD_WS=1
for V in 0 1; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks$V -- $R/scripts/hip_probe/conv_bench.bin wgrad 6 $V 20 3 52,128,256,3,1,32 13,512,1024,3,1,32 > $O/run$V.log 2>&1
  cp $O/ks$V/*/*kernel_stats.csv $O/kernel_stats_opt$V.csv; rm -rf $O/ks$V
done
cd $R
cp tf2_yolo_amd/libyolo_hip.so $O/prod.so
cp tf2_yolo_amd/libyolo_hip_ko.so.bin tf2_yolo_amd/libyolo_hip.so
cd /tmp
YOLO_WGRAD_KO=7 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks7 -- $R/scripts/hip_probe/conv_bench.bin wgrad 6 1 20 3 52,128,256,3,1,32 > $O/run7.log 2>&1
cp $O/ks7/*/*kernel_stats.csv $O/kernel_stats_ko7.csv; rm -rf $O/ks7
cd $R
cp $O/prod.so tf2_yolo_amd/libyolo_hip.so; rm $O/prod.so
head -8 $O/kernel_stats_opt0.csv $O/kernel_stats_opt1.csv $O/kernel_stats_ko7.csv | cut -c1-200
