#!/bin/bash
# round 6: fewer, longer workgroups for the filter gradients (x-window target 256 = one per CU; per-tap 3x3 target 512): all configs
mkdir -p gpurun_out/r6ac
run() {   # label, env assignments...
  echo "== $*" >> gpurun_out/r6ac/ab.log
  env "$@" python bench.py --plain --steps 20 --warmup 5 2>/dev/null | tail -n 1 >> gpurun_out/r6ac/ab.log
  env "$@" python scripts/bench_configs.py c1 c2 c4 2>/dev/null >> gpurun_out/r6ac/ab.log
}
run YOLO_WGRAD_WIN_TARGET=0
run YOLO_WGRAD_WIN_TARGET=256
run YOLO_WGRAD_WIN_TARGET=256 YOLO_WGRAD_TARGET=512
run YOLO_WGRAD_WIN_TARGET=0 YOLO_WGRAD_TARGET=512
run YOLO_WGRAD_WIN_TARGET=0
run YOLO_WGRAD_WIN_TARGET=256
cat gpurun_out/r6ac/ab.log
