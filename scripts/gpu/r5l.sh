#!/bin/bash
R=$PWD; O=$R/gpurun_out/r5l; mkdir -p $O
L="52,256,128,1,1,32 26,512,256,1,1,32 13,1024,512,1,1,32 104,128,64,1,1,32 52,384,128,1,1,32 26,768,256,1,1,32"
for D in 0 1 2; do for W in 0 4 8; do echo "== YOLO_PLANES_DEEP=$D YOLO_PLANES_WAVES=$W fwd"; YOLO_PLANES_DEEP=$D YOLO_PLANES_WAVES=$W timeout -k 5 120 scripts/hip_probe/conv_bench.bin fwd 0 1 20 3 $L 2>&1 | grep -i "us\|TF" | tail -8; done; done > $O/k1_fwd.log 2>&1
cat $O/k1_fwd.log | head -120
