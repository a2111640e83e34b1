#!/bin/bash
# round 6: kernel traces of C1 (after the small-launch changes), C2 and C5's bs-1 forward
mkdir -p gpurun_out/r6j
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in c1 c2; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r6j/$c -- python3 $R/scripts/bench_configs.py $c > $R/gpurun_out/r6j/$c.log 2>&1 || exit 1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r6j/c5 -- python3 $R/scripts/infer_bs1_graph.py > $R/gpurun_out/r6j/c5.log 2>&1
tail -2 $R/gpurun_out/r6j/c*.log
