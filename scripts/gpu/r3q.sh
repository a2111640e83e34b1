#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for R in 1.5e9 6e9 24e9; do
  echo "== ring $R"
  for l in 52,256,128,1 26,512,256,1; do
    COLD_RING_BYTES=$R timeout -k 10 300 python scripts/cold_probe.py $l 2>&1 | grep -v amdgpu || exit 1
  done
done 2>&1 | tee gpurun_out/r3q_cold.log
