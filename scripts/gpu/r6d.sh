#!/bin/bash
# knock-outs + per-kernel durations of the x-window filter gradient, slabs + ordered reduce (the production form)
O=$PWD/gpurun_out/r6d; mkdir -p $O; R=$PWD
export CONV_BENCH_WGRAD_WS=1
cd /tmp; export TMPDIR=/tmp
for V in 0 1; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks$V -- $R/scripts/hip_probe/conv_bench.bin wgrad 6 $V 20 3 52,128,256,3,1,32 13,512,1024,3,1,32 > $O/run$V.log 2>&1
  cp $O/ks$V/*/*kernel_stats.csv $O/kernel_stats_opt$V.csv; rm -rf $O/ks$V
done
cd $R
cp tf2_yolo_amd/libyolo_hip.so $O/prod.so
cp tf2_yolo_amd/libyolo_hip_ko.so.bin tf2_yolo_amd/libyolo_hip.so
for L in 52,128,256,3,1,32 13,512,1024,3,1,32; do
  for K in 0 1 2 3 4 7 8; do
    echo -n "win layer $L ko=$K: "; YOLO_WGRAD_KO=$K timeout -k 10 120 scripts/hip_probe/conv_bench.bin wgrad 6 1 20 3 $L 2>&1 | tail -1
  done
done 2>&1 | tee $O/wgrad_win_ko.log
cd /tmp
YOLO_WGRAD_KO=7 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks7 -- $R/scripts/hip_probe/conv_bench.bin wgrad 6 1 20 3 52,128,256,3,1,32 > $O/run7.log 2>&1
cp $O/ks7/*/*kernel_stats.csv $O/kernel_stats_ko7.csv; rm -rf $O/ks7
cd $R
cp $O/prod.so tf2_yolo_amd/libyolo_hip.so; rm $O/prod.so
head -5 $O/kernel_stats_opt0.csv $O/kernel_stats_opt1.csv $O/kernel_stats_ko7.csv | cut -c1-200
