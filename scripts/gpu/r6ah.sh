#!/bin/bash
# round 6: bs-1 predict with / without the one-launch kernel for units with few output pixels (conv_small.hip)
mkdir -p gpurun_out/r6ah
L=gpurun_out/r6ah/ab.log
for v in 0 1 0 1; do
  echo "== YOLO_CONV_SMALL=$v" >> $L
  YOLO_CONV_SMALL=$v python scripts/infer_bs1_graph.py 2>/dev/null >> $L
done
for g in 512 1024 4096; do
  echo "== YOLO_CONV_SMALL_GRID=$g" >> $L
  YOLO_CONV_SMALL_GRID=$g python scripts/infer_bs1_graph.py 2>/dev/null >> $L
done
cat $L
