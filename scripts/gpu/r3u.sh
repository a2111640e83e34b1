#!/bin/bash
O=gpurun_out/r3u; mkdir -p $O
python -m pytest tests/test_gpu_conv.py -x -q > $O/tests.log 2>&1; echo "tests rc $?"; tail -2 $O/tests.log
i=0
for V in "A=0" "YOLO_WGRAD_TARGET_1X1=512" "YOLO_WGRAD_TARGET_1X1=768" "YOLO_WGRAD_DEEP_1X1=5" "YOLO_WGRAD_DEEP_1X1=6" "YOLO_WGRAD_TARGET_1X1=512 YOLO_WGRAD_DEEP_1X1=5" "YOLO_WGRAD_TARGET_1X1=512 YOLO_WGRAD_DEEP_1X1=4" "A=0"; do
  i=$((i+1))
  env $V python bench.py --no-cpu-baseline --no-kernel-timer --steps 20 > $O/bench_$i.log 2>$O/bench_$i.err || { tail -5 $O/bench_$i.err; exit 1; }
  echo -n "$V: "; python scripts/bench_line.py $O/bench_$i.log
  env $V python scripts/instep_1x1.py 2>&1 | grep "wgrad.* 1 1 " 
done
