#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_conv.py -q -x -k "wgrad" > gpurun_out/r3n_tests.log 2>&1; echo "tests rc $?"; tail -3 gpurun_out/r3n_tests.log
timeout -k 10 300 python scripts/instep_probe.py 2>&1 | tee gpurun_out/r3n_probe.log
