#!/bin/bash
# round 4, step a: the x-window filter gradient (conv_wgrad_win.hip): parity tests, then A/B against the per-tap kernel
set -o pipefail
O=gpurun_out/r6a; mkdir -p $O
hipcc -O2 -std=c++17 scripts/hip_probe/conv_bench.cpp -Iinclude -Ltf2_yolo_amd -lyolo_hip -Wl,-rpath,'$ORIGIN/../../tf2_yolo_amd' -o scripts/hip_probe/conv_bench.bin > $O/build.log 2>&1 || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_conv.py -x -q -k "wgrad" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -5 $O/tests.log
L="52,128,256,3,1,32 26,256,512,3,1,32 13,512,1024,3,1,32"
timeout -k 10 300 scripts/hip_probe/conv_bench.bin wgrad 6 0,1 20 5 $L > $O/ab_atomics.log 2>&1
CONV_BENCH_WGRAD_WS=1 timeout -k 10 300 scripts/hip_probe/conv_bench.bin wgrad 6 0,1 20 5 $L > $O/ab_slabs.log 2>&1
cat $O/ab_atomics.log $O/ab_slabs.log
