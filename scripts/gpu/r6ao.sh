#!/bin/bash
# is the folded BatchNorm-backward reduction reproducible run to run? (C2, 11 steps, loss printed to 1e-4)
mkdir -p gpurun_out/r6ao
L=gpurun_out/r6ao/ab.log
for i in 1 2 3 4 5 6; do
  YOLO_BN_FOLD=1 python scripts/bench_configs.py c2 2>/dev/null >> $L
done
for i in 1 2 3; do
  YOLO_BN_FOLD=0 python scripts/bench_configs.py c2 2>/dev/null >> $L
done
cut -c90-200 $L
