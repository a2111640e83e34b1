#!/bin/bash
# round 6: one-split filter gradients add to dw themselves; bn_finalize with coalesced slot reads: tests + C1 / C2 A/B
mkdir -p gpurun_out/r6i
python -m pytest tests/test_gpu_conv.py tests/test_gpu_elementwise.py tests/test_gpu_loss.py tests/test_gpu_kats.py -x -q > gpurun_out/r6i/tests_kernels.log 2>&1; tail -3 gpurun_out/r6i/tests_kernels.log
python -m pytest tests/test_gpu_model.py tests/test_gpu_keras_shell.py -x -q -k "not bs32 and not 608" > gpurun_out/r6i/tests_model.log 2>&1; tail -3 gpurun_out/r6i/tests_model.log
for exp in 48 32 0 48 0; do
  echo "== YOLO_EXP=$exp" >> gpurun_out/r6i/ab.log
  YOLO_EXP=$exp python scripts/bench_configs.py c1 c2 2>/dev/null >> gpurun_out/r6i/ab.log || exit 1
done
cat gpurun_out/r6i/ab.log
