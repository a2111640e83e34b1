#!/bin/bash
R=$PWD; O=$R/gpurun_out/r5a; mkdir -p $O; export TMPDIR=/tmp
python -m pytest tests/test_gpu_conv.py -x -q -k "fused_bn_backward" > $O/t1.log 2>&1; echo "bnred kernel tests rc $?"; tail -8 $O/t1.log
python -m pytest tests/test_gpu_model.py tests/test_gpu_keras_shell.py -x -q -k "not 608 and not 416" > $O/t2.log 2>&1; echo "model tests rc $?"; tail -5 $O/t2.log
python scripts/step_ab.py --k 10 --rounds 3 "off:YOLO_BN_FUSED_REDUCE=0" "on:YOLO_BN_FUSED_REDUCE=1" > $O/ab_c3.log 2>&1; echo "ab rc $?"; cat $O/ab_c3.log | tail -8
