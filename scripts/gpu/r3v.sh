#!/bin/bash
O=gpurun_out/r3v; mkdir -p $O
for V in "YOLO_PLANES_DEEP=0" "A=0" "YOLO_PLANES_DEEP=0" "A=0"; do
  echo -n "$V: "; env $V python scripts/bench_configs.py c4 2>&1 | grep images_per_s
done
python scripts/bench_configs.py c1 c2 c5 2>&1 | grep -v amdgpu
