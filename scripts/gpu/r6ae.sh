#!/bin/bash
# round 6: x-window filter gradient workgroups: 64 .. 512
mkdir -p gpurun_out/r6ae
for t in 512 64 96 128 160 256 512 128; do
  echo "== YOLO_WGRAD_WIN_TARGET=$t" >> gpurun_out/r6ae/ab.log
  YOLO_WGRAD_WIN_TARGET=$t python bench.py --plain --steps 20 --warmup 5 2>/dev/null | tail -n 1 >> gpurun_out/r6ae/ab.log
  YOLO_WGRAD_WIN_TARGET=$t python scripts/bench_configs.py c2 c4 2>/dev/null >> gpurun_out/r6ae/ab.log
done
cat gpurun_out/r6ae/ab.log
