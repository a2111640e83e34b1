#!/bin/bash
O=gpurun_out/r3y; mkdir -p $O
python -m pytest tests/test_gpu_conv.py -x -q -k "inference_unit or epilogue or split_k" > $O/t1.log 2>&1; echo "conv tests rc $?"; tail -12 $O/t1.log
python -m pytest tests/test_gpu_model.py tests/test_gpu_keras_shell.py -x -q -k "not 608 and not 416" > $O/t2.log 2>&1; echo "model tests rc $?"; tail -5 $O/t2.log
for V in 0 1 0 1; do echo "onepass=$V"; YOLO_INFER_ONEPASS=$V python scripts/bench_configs.py c5 2>&1 | grep "inference forward"; done
