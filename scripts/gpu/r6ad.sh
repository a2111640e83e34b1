#!/bin/bash
# round 6: x-window filter gradient workgroups below one per CU
mkdir -p gpurun_out/r6ad
for t in 0 192 128 0; do
  echo "== YOLO_WGRAD_WIN_TARGET=$t (0 = the new default, one per CU)" >> gpurun_out/r6ad/ab.log
  YOLO_WGRAD_WIN_TARGET=$t python bench.py --plain --steps 20 --warmup 5 2>/dev/null | tail -n 1 >> gpurun_out/r6ad/ab.log
  YOLO_WGRAD_WIN_TARGET=$t python scripts/bench_configs.py c2 c4 2>/dev/null >> gpurun_out/r6ad/ab.log
done
cat gpurun_out/r6ad/ab.log
timeout -k 10 300 python -m pytest tests/test_gpu_conv.py -x -q -k "wgrad" > gpurun_out/r6ad/tests.log 2>&1; tail -2 gpurun_out/r6ad/tests.log
