#!/bin/bash
O=gpurun_out/r5f; mkdir -p $O
for V in 1 2 0 1 2 0; do
  YOLO_BN_TIGHT_BOUND=$V python bench.py --no-cpu-baseline --no-kernel-timer --steps 20 > $O/c.log 2>$O/c.err; echo -n "tight=$V: "; python scripts/bench_line.py $O/c.log
done
YOLO_BN_TIGHT_BOUND=0 python -m pytest tests/test_gpu_model.py -x -q -k "not 608 and not 416" > $O/t.log 2>&1; echo "tests(tight=0) rc $?"; tail -3 $O/t.log
