#!/bin/bash
# round 6: kernel trace of the C4 step (YOLOv4-608 bs 16), two streams
mkdir -p gpurun_out/r6l
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r6l/c4 -- python3 $R/scripts/bench_configs.py c4 > $R/gpurun_out/r6l/c4.log 2>&1 || exit 1
tail -n 2 $R/gpurun_out/r6l/c4.log
