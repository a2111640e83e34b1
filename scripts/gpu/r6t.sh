#!/bin/bash
# round 6: x-window filter gradient with two pixel halves per workgroup (half the slabs): tests, standalone layers, steps
mkdir -p gpurun_out/r6t
timeout -k 10 300 python -m pytest tests/test_gpu_conv.py -x -q -k "wgrad_window" > gpurun_out/r6t/tests.log 2>&1; tail -3 gpurun_out/r6t/tests.log
grep -q "passed" gpurun_out/r6t/tests.log || exit 1
grep -q "failed" gpurun_out/r6t/tests.log && exit 1
for v in 1 4 1 4; do
  echo "== YOLO_WGRAD_WIN=$v" >> gpurun_out/r6t/ab.log
  YOLO_WGRAD_WIN=$v timeout -k 10 200 python bench.py --plain --steps 20 --warmup 5 2>/dev/null | tail -n 1 >> gpurun_out/r6t/ab.log || exit 1
  YOLO_WGRAD_WIN=$v timeout -k 10 200 python scripts/bench_configs.py c2 c4 2>/dev/null >> gpurun_out/r6t/ab.log || exit 1
done
cat gpurun_out/r6t/ab.log
