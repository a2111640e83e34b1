# usage: bash scripts/win_sweep.sh <tag>   (GPU box) -- window-kernel A/B on the 3x3 stride-1 layers of YOLOv3-416 bs 32
T=${1:-w}
B=scripts/hip_probe/conv_bench.bin
L="208,32,64,3,1,32 104,64,128,3,1,32 52,128,256,3,1,32 26,256,512,3,1,32 13,512,1024,3,1,32"
timeout -k 10 300 $B fwd 0 0,2,4 10 7 $L > gpurun_out/${T}_fwd.log 2>&1 && \
timeout -k 10 300 $B dgrad 0 0,2,4 10 7 $L > gpurun_out/${T}_dgrad.log 2>&1
