#!/bin/bash
mkdir -p gpurun_out/r6al
L=gpurun_out/r6al/ab.log
for v in 2 1 2 1; do
  echo "== YOLO_WIN_SPLIT_MIN_CB=$v" >> $L
  YOLO_WIN_SPLIT_MIN_CB=$v python scripts/infer_bs1_graph.py 2>/dev/null >> $L
done
cat $L
