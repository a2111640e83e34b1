#!/bin/bash
# concat gradients read in place by the BatchNorm backward: tests, then C4 / C3 A/B
O=gpurun_out/r6x; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_elementwise.py tests/test_gpu_model.py -x -q -k "channel_slice or test_bn_act_train or 4-True-False or 3-True-False or 2-True-False" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log; tail -3 $O/tests.log
for i in 1 2; do for V in 0 1; do
  echo -n "C4 CONCAT_GRAD_SLICE=$V run $i: "; YOLO_CONCAT_GRAD_SLICE=$V python scripts/bench_configs.py c4 2>/dev/null | cut -c60-130
done; done
for V in 0 1; do echo -n "C3 CONCAT_GRAD_SLICE=$V: "; YOLO_CONCAT_GRAD_SLICE=$V python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --no-extra-blocks 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'])"; done
