#!/bin/bash
O=gpurun_out/r3s; mkdir -p $O
python -m pytest tests/test_gpu_conv.py -x -q -k "planes or c3 or C3" > $O/tests.log 2>&1; echo "tests rc $?"; tail -2 $O/tests.log
i=0
for V in "A=0" "YOLO_PLANES_WAVES=4" "YOLO_PLANES_DEEP=1" "YOLO_PLANES_WAVES=4 YOLO_PLANES_DEEP=1" "A=0" "YOLO_PLANES_DEEP=1"; do
  i=$((i+1))
  env $V python bench.py --no-cpu-baseline --no-kernel-timer --steps 20 > $O/bench_$i.log 2>$O/bench_$i.err || { tail -5 $O/bench_$i.err; exit 1; }
  echo -n "$V: "; python scripts/bench_line.py $O/bench_$i.log
done
for V in "A=0" "YOLO_PLANES_DEEP=1" "YOLO_PLANES_WAVES=4 YOLO_PLANES_DEEP=1"; do
  echo "== $V"; env $V python scripts/instep_1x1.py 2>&1 | grep " 1 1 " 
done
