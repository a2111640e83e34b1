#!/bin/bash
# x-window filter gradient: prefetch distance 1 vs 2, per-tap kernel beside them; random and zero data (power or latency?)
O=$PWD/gpurun_out/r6e; mkdir -p $O
export CONV_BENCH_WGRAD_WS=1
L="52,128,256,3,1,32 26,256,512,3,1,32 13,512,1024,3,1,32"
timeout -k 10 300 scripts/hip_probe/conv_bench.bin wgrad 6 0,1,2 20 5 $L > $O/ab.log 2>&1
CONV_BENCH_ZEROS=1 timeout -k 10 300 scripts/hip_probe/conv_bench.bin wgrad 6 0,1,2 20 5 $L > $O/ab_zeros.log 2>&1
cat $O/ab.log $O/ab_zeros.log
