#!/bin/bash
# round-3 GPU batch f: launch tape vs hipGraph vs eager (tests, host time, step rate on one box)
O=gpurun_out/r3f; mkdir -p $O
python -m pytest tests/test_gpu_keras_shell.py tests/test_gpu_dp.py -x -q > $O/tests.log 2>&1; echo "tests rc $?"; grep -v "frame #" $O/tests.log | tail -12
for M in tape graph eager; do YOLO_STEP_MODE=$M python scripts/host_enqueue.py 10 2>&1 | grep -v amdgpu.ids | tee $O/host_$M.log; done
for M in tape eager tape graph; do YOLO_STEP_MODE=$M python bench.py --no-cpu-baseline > $O/bench_$M.log 2>$O/bench_$M.err; python scripts/bench_line.py $O/bench_$M.log; done
