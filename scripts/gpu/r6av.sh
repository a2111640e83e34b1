#!/bin/bash
# inference tests under every switch of the bs-1 path
mkdir -p gpurun_out/r6av
for e in "YOLO_INFER_ONEPASS=0" "YOLO_INFER_FUSE=0" "YOLO_CONV_SMALL=0" "YOLO_INFER_SMALL_FUSE=0" "YOLO_CONCAT_PLANES=0" "YOLO_CONV_SMALL=3" "YOLO_CONV_SMALL=0 YOLO_INFER_SMALL_FUSE=0"; do
  env $e timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "infer or predict or keras_shell or readme" > gpurun_out/r6av/t.log 2>&1 || { echo "FAILED under $e"; tail -30 gpurun_out/r6av/t.log; exit 1; }
  echo "$e: $(tail -1 gpurun_out/r6av/t.log)"
done
