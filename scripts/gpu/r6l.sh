#!/bin/bash
# the x-window filter gradient on the 16x16x32 MFMA: parity, then A/B (0 per-tap, 1 window 32x32x16, 3 window 16x16x32)
O=gpurun_out/r6l; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_conv.py -x -q -k "window_kernel" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -4 $O/tests.log
export CONV_BENCH_WGRAD_WS=1
L="52,128,256,3,1,32 26,256,512,3,1,32 13,512,1024,3,1,32"
timeout -k 10 300 scripts/hip_probe/conv_bench.bin wgrad 6 0,1,3 20 5 $L > $O/ab.log 2>&1
CONV_BENCH_ZEROS=1 timeout -k 10 300 scripts/hip_probe/conv_bench.bin wgrad 6 1,3 20 5 52,128,256,3,1,32 > $O/ab_zeros.log 2>&1
cat $O/ab.log $O/ab_zeros.log
