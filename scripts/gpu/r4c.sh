#!/bin/bash
O=gpurun_out/r4c; mkdir -p $O
python -m pytest tests/test_gpu_conv.py -x -q -k "inference_unit" > $O/t1.log 2>&1; echo "conv tests rc $?"; tail -5 $O/t1.log
