#!/bin/bash
# round 6: conv_small.hip with its launch policy: tests (default, 3x3 units too, forced tiles), then bs-1 predict A/B
mkdir -p gpurun_out/r6aj
L=gpurun_out/r6aj/ab.log
for e in "YOLO_NOP=1" "YOLO_CONV_SMALL=3" "YOLO_CONV_SMALL=3 YOLO_CONV_SMALL_TILE=21" "YOLO_CONV_SMALL=3 YOLO_CONV_SMALL_TILE=22" "YOLO_CONV_SMALL=0"; do
  env $e timeout -k 10 300 python -m pytest tests/test_gpu_conv.py -x -q -k "inference_unit" > gpurun_out/r6aj/test.log 2>&1 || { echo "FAILED under $e"; tail -30 gpurun_out/r6aj/test.log; exit 1; }
  echo "$e: $(tail -1 gpurun_out/r6aj/test.log)"
done
for v in 0 1 0 1; do
  echo "== YOLO_CONV_SMALL=$v" >> $L
  YOLO_CONV_SMALL=$v python scripts/infer_bs1_graph.py 2>/dev/null >> $L
done
cat $L
