#!/bin/bash
# round 6: kernel trace of C2 after the pool fusion
mkdir -p gpurun_out/r6x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r6x/c2 -- python3 $R/scripts/bench_configs.py c2 > $R/gpurun_out/r6x/c2.log 2>&1 || exit 1
tail -n 1 $R/gpurun_out/r6x/c2.log
