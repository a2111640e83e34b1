#!/bin/bash
R=$PWD; O=$R/gpurun_out/r5d; mkdir -p $O; export TMPDIR=/tmp
python -m pytest tests/test_gpu_conv.py -x -q -k "fused_bn_backward or wgrad_window or wgrad_planes" > $O/t1.log 2>&1; echo "conv tests rc $?"; tail -3 $O/t1.log
python -m pytest tests/test_gpu_model.py -x -q -k "fused_bn" > $O/t2.log 2>&1; echo "fused model test rc $?"; tail -5 $O/t2.log
python -m pytest tests/test_gpu_elementwise.py tests/test_gpu_keras_shell.py -x -q > $O/t3.log 2>&1; echo "elementwise+shell rc $?"; tail -3 $O/t3.log
python -m pytest tests/test_gpu_dp.py -x -q -s > $O/t4.log 2>&1; echo "dp rc $?"; tail -5 $O/t4.log
