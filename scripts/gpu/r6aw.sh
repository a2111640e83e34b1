#!/bin/bash
# the inference-unit tests (fixed cases + 24 random shapes) under the default policy and with every 3x3 unit through conv_small.hip
mkdir -p gpurun_out/r6aw
for e in "YOLO_NOP=1" "YOLO_CONV_SMALL=3" "YOLO_CONV_SMALL=3 YOLO_CONV_SMALL_TILE=21" "YOLO_CONV_SMALL=3 YOLO_CONV_SMALL_TILE=22"; do
  env $e timeout -k 10 300 python -m pytest tests/test_gpu_conv.py -x -q -k "inference_unit or random_shapes or head_unit" > gpurun_out/r6aw/t.log 2>&1 || { echo "FAILED under $e"; tail -40 gpurun_out/r6aw/t.log; exit 1; }
  echo "$e: $(tail -1 gpurun_out/r6aw/t.log)"
done
