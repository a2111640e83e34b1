#!/bin/bash
# round 6: the launch-bound small configurations with the BatchNorm-backward reduction finished by its own launch (72+ launches fewer)
mkdir -p gpurun_out/r6an
L=gpurun_out/r6an/ab.log
for v in 0 1 0 1; do
  echo "== YOLO_BN_FOLD=$v" >> $L
  YOLO_BN_FOLD=$v python scripts/bench_configs.py c1 c2 2>/dev/null >> $L
done
cat $L
