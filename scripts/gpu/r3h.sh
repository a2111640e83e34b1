#!/bin/bash
O=gpurun_out/r3h; mkdir -p $O
python scripts/cold_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/cold.log
for M in tape eager; do YOLO_STEP_MODE=$M python scripts/bench_configs.py c4 2>&1 | grep -v amdgpu.ids | tee -a $O/c4_modes.log; done
