#!/bin/bash
# 1x1 filter gradients: how far from their byte floor? (C3 and C4 shapes; kernel + ordered reduce; fwd twin beside)
export CONV_BENCH_WGRAD_WS=1
L3="52,256,128,1,1,32 26,512,256,1,1,32 13,1024,512,1,1,32 104,128,64,1,1,32"
L4="76,256,128,1,1,16 76,128,128,1,1,16 38,512,256,1,1,16 38,256,256,1,1,16 152,128,64,1,1,16 152,64,64,1,1,16 19,1024,512,1,1,16"
scripts/hip_probe/conv_bench.bin wgrad 6 1 20 5 $L3 $L4 2>&1 | grep -E "wgrad H|opt6"
scripts/hip_probe/conv_bench.bin fwd 0 1 20 5 $L3 $L4 2>&1 | grep -E "fwd H|opt0"
