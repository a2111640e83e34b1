#!/bin/bash
O=gpurun_out/r3t; mkdir -p $O
i=0
for V in "YOLO_PLANES_DEEP=0" "YOLO_PLANES_DEEP=1" "YOLO_PLANES_DEEP=2" "YOLO_PLANES_DEEP=0" "YOLO_PLANES_DEEP=1" "YOLO_PLANES_DEEP=2"; do
  i=$((i+1))
  env $V python bench.py --no-cpu-baseline --no-kernel-timer --steps 20 > $O/bench_$i.log 2>$O/bench_$i.err || { tail -5 $O/bench_$i.err; exit 1; }
  echo -n "$V: "; python scripts/bench_line.py $O/bench_$i.log
done
for V in "YOLO_PLANES_DEEP=1" "YOLO_PLANES_DEEP=2"; do
  echo "== $V"; env $V python scripts/instep_1x1.py 2>&1 | grep " 1 1 " 
done
