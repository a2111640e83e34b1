#!/bin/bash
# knock-outs of the x-window filter gradient (diagnostic build swapped in for the run; results are wrong by construction)
O=gpurun_out/r6b; mkdir -p $O
cp tf2_yolo_amd/libyolo_hip.so $O/prod.so
cp tf2_yolo_amd/libyolo_hip_ko.so.bin tf2_yolo_amd/libyolo_hip.so
export CONV_BENCH_WGRAD_WS=1
for L in 52,128,256,3,1,32 13,512,1024,3,1,32; do
  for K in 0 1 2 3 4 7 8; do
    echo -n "win layer $L ko=$K: "; YOLO_WGRAD_KO=$K timeout -k 10 120 scripts/hip_probe/conv_bench.bin wgrad 6 1 20 3 $L 2>&1 | tail -1
  done
done 2>&1 | tee $O/wgrad_win_ko.log
cp $O/prod.so tf2_yolo_amd/libyolo_hip.so; rm $O/prod.so
