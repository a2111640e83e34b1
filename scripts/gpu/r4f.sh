#!/bin/bash
# knock-out experiment on the filter-gradient kernel (diagnostic build of the library swapped in for the run).
# Before: make KNOCKOUTS=1 -B -j8 && cp tf2_yolo_amd/libyolo_hip.so tf2_yolo_amd/libyolo_hip_ko.so.bin && make -B -j8
O=gpurun_out/r4f; mkdir -p $O
cp tf2_yolo_amd/libyolo_hip.so $O/prod.so
cp tf2_yolo_amd/libyolo_hip_ko.so.bin tf2_yolo_amd/libyolo_hip.so
for L in 52,128,256,3,1,32 26,256,512,3,1,32 13,512,1024,3,1,32; do
  for K in 0 1 2 3 4 7; do
    echo -n "layer $L ko=$K: "; YOLO_WGRAD_KO=$K timeout -k 10 120 scripts/hip_probe/conv_bench.bin wgrad 0 1 20 3 $L 2>&1 | tail -1
  done
done 2>&1 | tee $O/wgrad_ko.log
cp $O/prod.so tf2_yolo_amd/libyolo_hip.so; rm $O/prod.so
