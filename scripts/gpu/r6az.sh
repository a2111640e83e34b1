#!/bin/bash
mkdir -p gpurun_out/r6az
L=gpurun_out/r6az/ab.log
for v in 0 1 0 1; do
  echo "== YOLO_CONV_SMALL_2R=$v" >> $L
  YOLO_CONV_SMALL_2R=$v python scripts/infer_bs1_graph.py 2>/dev/null >> $L
done
cat $L
