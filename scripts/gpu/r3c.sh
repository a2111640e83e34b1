#!/bin/bash
# round-3 GPU batch c: full GPU suite, then same-box A/B of the captured step and of the atomics-free filter gradients
set -o pipefail
O=gpurun_out/r3c; mkdir -p $O
python -m pytest tests -m gpu -x -q --durations=10 > $O/tests.log 2>&1; echo "tests rc $?"; tail -16 $O/tests.log
python bench.py --no-cpu-baseline > $O/bench_graph.log 2>$O/bench_graph.err || { tail -5 $O/bench_graph.err; exit 1; }
YOLO_STEP_GRAPH=0 python bench.py --no-cpu-baseline > $O/bench_eager.log 2>$O/bench_eager.err || exit 1
YOLO_WGRAD_DETERMINISTIC=0 python bench.py --no-cpu-baseline > $O/bench_atomics.log 2>$O/bench_atomics.err || exit 1
python bench.py --no-cpu-baseline > $O/bench_graph2.log 2>$O/bench_graph2.err || exit 1
for f in $O/bench_*.log; do python scripts/bench_line.py $f; done
grep -h "eager + per-launch" $O/bench_*.err
