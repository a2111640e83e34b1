#!/bin/bash
# round 6: fused heads, concat reading through the upsampling, lateral conv as a one-pass unit: tests, then bs-1 predict A/B
mkdir -p gpurun_out/r6am
timeout -k 10 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_elementwise.py -x -q -k "head_unit or concat or inference_unit" > gpurun_out/r6am/t1.log 2>&1 || { tail -40 gpurun_out/r6am/t1.log; exit 1; }
tail -1 gpurun_out/r6am/t1.log
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "infer or predict or keras_shell or readme or parity" > gpurun_out/r6am/t2.log 2>&1 || { tail -40 gpurun_out/r6am/t2.log; exit 1; }
tail -1 gpurun_out/r6am/t2.log
L=gpurun_out/r6am/ab.log
for v in 0 1 0 1; do
  echo "== YOLO_INFER_SMALL_FUSE=$v" >> $L
  YOLO_INFER_SMALL_FUSE=$v python scripts/infer_bs1_graph.py 2>/dev/null >> $L
done
cat $L
