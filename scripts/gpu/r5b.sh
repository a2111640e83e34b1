#!/bin/bash
O=gpurun_out/r5b; mkdir -p $O
python -m pytest tests/test_gpu_conv.py tests/test_gpu_model.py tests/test_gpu_keras_shell.py -x -q -k "not 608 and not 416" > $O/t.log 2>&1; echo "tests rc $?"; tail -3 $O/t.log
python scripts/bench_configs.py c1 c2 c5 2>&1 | grep -v amdgpu | head -5 | cut -c1-170
