#!/bin/bash
# round 6: C1 / C2 with the in-launch BatchNorm fold on and off, tape and hipGraph replay; a kernel trace of C1
mkdir -p gpurun_out/r6g
for fold in 0 1; do for mode in tape graph; do
  echo "== YOLO_BN_FOLD=$fold YOLO_STEP_MODE=$mode" >> gpurun_out/r6g/c1c2.log
  YOLO_BN_FOLD=$fold YOLO_STEP_MODE=$mode python scripts/bench_configs.py c1 c2 2>/dev/null >> gpurun_out/r6g/c1c2.log || exit 1
done; done
cat gpurun_out/r6g/c1c2.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r6g/c1prof -- python3 $GRAFT_REPO_ROOT/scripts/bench_configs.py c1 > $GRAFT_REPO_ROOT/gpurun_out/r6g/c1prof.log 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/r6g/c1prof/*/ | head
