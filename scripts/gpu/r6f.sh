#!/bin/bash
# round 4: full GPU suite with the x-window filter gradient as the default, then the step A/B (YOLO_WGRAD_WIN=0/1, two alternating pairs)
O=gpurun_out/r6f; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log
tail -3 $O/tests.log
for i in 1 2; do
  for V in 0 1; do
    YOLO_WGRAD_WIN=$V timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer 2> $O/bench_${V}_$i.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('WGRAD_WIN=$V run $i:', j['value'], 'img/s', j['ms_per_step'], 'ms')" | tee -a $O/ab.log
  done
done
