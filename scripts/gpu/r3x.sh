#!/bin/bash
O=gpurun_out/r3x; mkdir -p $O
python -m pytest tests/test_gpu_conv.py tests/test_gpu_model.py -x -q -k "not 608" > $O/tests.log 2>&1; echo "tests rc $?"; tail -3 $O/tests.log
i=0
for V in "YOLO_PLANES_DEEP=0" "A=0" "YOLO_PLANES_DEEP=1" "YOLO_PLANES_DEEP=0" "A=0" "YOLO_PLANES_DEEP=1"; do
  i=$((i+1))
  env $V python bench.py --no-cpu-baseline --no-kernel-timer --steps 20 > $O/bench_$i.log 2>$O/bench_$i.err || { tail -5 $O/bench_$i.err; exit 1; }
  echo -n "$V: "; python scripts/bench_line.py $O/bench_$i.log
done
for V in "YOLO_PLANES_DEEP=0" "A=0" "YOLO_PLANES_DEEP=0" "A=0"; do
  echo -n "$V: "; env $V python scripts/bench_configs.py c4 2>&1 | grep images_per_s | cut -c1-200
done
