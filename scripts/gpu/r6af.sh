#!/bin/bash
# round 6: with the x-window filter gradient at 128 workgroups -- fewer workgroups for the per-tap filter gradients too?
mkdir -p gpurun_out/r6af
run() {
  echo "== $*" >> gpurun_out/r6af/ab.log
  env "$@" python bench.py --plain --steps 20 --warmup 5 2>/dev/null | tail -n 1 >> gpurun_out/r6af/ab.log
  env "$@" python scripts/bench_configs.py c2 c4 2>/dev/null >> gpurun_out/r6af/ab.log
}
run YOLO_NOP=1
run YOLO_WGRAD_TARGET_1X1=128
run YOLO_WGRAD_TARGET_1X1=64
run YOLO_WGRAD_TARGET=256
run YOLO_WGRAD_TARGET=384
run YOLO_WGRAD_TARGET_1X1=128 YOLO_WGRAD_TARGET=256
run YOLO_NOP=1
cat gpurun_out/r6af/ab.log
