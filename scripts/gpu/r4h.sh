#!/bin/bash
O=gpurun_out/r4h; mkdir -p $O
i=0
for V in 0 2 0 2 1; do
  i=$((i+1))
  YOLO_WGRAD_WIDE=$V python bench.py --no-cpu-baseline --no-kernel-timer --steps 20 > $O/bench_$i.log 2>$O/bench_$i.err || { tail -5 $O/bench_$i.err; exit 1; }
  echo -n "wide=$V: "; python scripts/bench_line.py $O/bench_$i.log
done
