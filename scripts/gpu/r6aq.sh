#!/bin/bash
# the pinned-row race of Adam's captured form (fixed in round 6): the test, then YOLOv2-416 parameters after 11 steps with the
# host running ahead (REPRO_SYNC=0) and synchronised per step -- one sha1 everywhere
mkdir -p gpurun_out/r6aq
timeout -k 10 300 python -m pytest tests/test_gpu_keras_shell.py -x -q -k "runs_ahead or captured_step" > gpurun_out/r6aq/new.log 2>&1; echo "$(tail -1 gpurun_out/r6aq/new.log)"
for i in 1 2 3; do REPRO_SYNC=0 timeout -k 10 200 python scripts/step_repro.py c2 11 2>&1 | grep "sha1" >> gpurun_out/r6aq/repro.log; done
REPRO_SYNC=1 timeout -k 10 200 python scripts/step_repro.py c2 11 2>&1 | grep "sha1" >> gpurun_out/r6aq/repro.log
cat gpurun_out/r6aq/repro.log
