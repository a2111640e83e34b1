#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python scripts/instep_1x1.py 2>&1 | grep -v amdgpu | tee gpurun_out/r3r_instep.log
