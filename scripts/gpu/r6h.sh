#!/bin/bash
# round 6: forward launches with BatchNorm statistics may split (conv_split_reduce_kernel makes the statistics): tests + A/B
mkdir -p gpurun_out/r6h
python -m pytest tests/test_gpu_conv.py tests/test_gpu_elementwise.py tests/test_gpu_loss.py -x -q > gpurun_out/r6h/tests_kernels.log 2>&1; tail -3 gpurun_out/r6h/tests_kernels.log
python -m pytest tests/test_gpu_model.py tests/test_gpu_keras_shell.py -x -q -k "not bs32" > gpurun_out/r6h/tests_model.log 2>&1; tail -3 gpurun_out/r6h/tests_model.log
for exp in 16 0 16 0; do
  echo "== YOLO_EXP=$exp" >> gpurun_out/r6h/ab.log
  YOLO_EXP=$exp python scripts/bench_configs.py c1 c2 c4 2>/dev/null >> gpurun_out/r6h/ab.log || exit 1
done
cat gpurun_out/r6h/ab.log
