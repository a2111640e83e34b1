#!/bin/bash
O=gpurun_out/r4g; mkdir -p $O
YOLO_WGRAD_WIDE=1 python -m pytest tests/test_gpu_conv.py -x -q -k "wgrad" > $O/t.log 2>&1; echo "tests rc $?"; tail -3 $O/t.log
for L in 52,128,256,3,1,32 26,256,512,3,1,32 13,512,1024,3,1,32 104,64,128,3,1,32 52,256,128,1,1,32 26,512,256,1,1,32; do
  for W in 0 1; do
    echo -n "layer $L wide=$W: "; YOLO_WGRAD_WIDE=$W timeout -k 10 120 scripts/hip_probe/conv_bench.bin wgrad 0 1 20 3 $L 2>&1 | tail -1
  done
done 2>&1 | tee $O/wgrad_wide.log
