#!/bin/bash
# round 6: timelines of the C3 step with the BatchNorm-backward fold off and on (why the 72 launches it removes buy nothing)
mkdir -p gpurun_out/r6aa
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for f in 0 1; do
  export YOLO_BN_FOLD=$f
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r6aa/fold$f -- python3 $R/bench.py --plain --steps 4 --warmup 3 > $R/gpurun_out/r6aa/fold$f.log 2>&1 || exit 1
  python3 $R/scripts/step_timeline.py $R/gpurun_out/r6aa/fold$f 2 > $R/gpurun_out/r6aa/timeline_fold$f.txt
  rm -rf $R/gpurun_out/r6aa/fold$f
done
head -12 $R/gpurun_out/r6aa/timeline_fold0.txt; head -12 $R/gpurun_out/r6aa/timeline_fold1.txt
