#!/bin/bash
# round 6 (needs scripts/experiments/wgrad_1x1_inline.patch applied): the 1x1 filter gradients on the COMPUTE stream (where a fused "1x1 backward" -- data + filter gradient from one
# read of dy, VERDICT r05 next #1 -- would have to run) against the side stream; same box, A B A
mkdir -p gpurun_out/r6ag
run() {
  echo "== $*" >> gpurun_out/r6ag/ab.log
  env "$@" python bench.py --plain --steps 20 --warmup 5 2>/dev/null | tail -n 1 >> gpurun_out/r6ag/ab.log
  env "$@" python scripts/bench_configs.py c4 2>/dev/null >> gpurun_out/r6ag/ab.log
}
run YOLO_NOP=1
run YOLO_WGRAD_1X1_INLINE=1
run YOLO_NOP=1
run YOLO_WGRAD_1X1_INLINE=1
cat gpurun_out/r6ag/ab.log
