#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for v in "" "YOLO_PLANES_WAVES=4" "YOLO_PLANES_NARROW_BELOW=100000"; do
  echo "== variant: $v"
  for l in 52,256,128,1 26,512,256,1 13,1024,512,1; do
    env $v timeout -k 10 200 python scripts/cold_probe.py $l 2>&1 | grep "conv fwd" || exit 1
  done
done 2>&1 | tee gpurun_out/r3o_cold.log
