#!/bin/bash
O=gpurun_out/r4q; mkdir -p $O
python -m pytest tests/test_gpu_elementwise.py tests/test_gpu_kats.py -x -q > $O/t1.log 2>&1; echo "tests rc $?"; tail -4 $O/t1.log
python -m pytest tests/test_gpu_model.py -x -q -k "4-" > $O/t2.log 2>&1; echo "v4 model tests rc $?"; tail -3 $O/t2.log
for V in 0 1 0 1; do echo -n "pool_plane=$V: "; YOLO_POOL_PLANE=$V python scripts/bench_configs.py c4 2>&1 | grep images_per_s | cut -c1-170; done
