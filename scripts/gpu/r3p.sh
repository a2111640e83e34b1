#!/bin/bash
O=gpurun_out/r3p; mkdir -p $O
python -m pytest tests/test_gpu_elementwise.py -x -q > $O/tests.log 2>&1; echo "tests rc $?"; tail -2 $O/tests.log
for V in 0 1 2 4 3 6 5 7 0; do
  YOLO_BN_REVERSE=$V python bench.py --no-cpu-baseline --no-kernel-timer --steps 20 > $O/bench_$V.log 2>$O/bench_$V.err || { tail -5 $O/bench_$V.err; exit 1; }
  echo -n "rev=$V "; python scripts/bench_line.py $O/bench_$V.log
done
