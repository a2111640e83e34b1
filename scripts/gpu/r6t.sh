#!/bin/bash
# hipGraph replay of the step, now that streams have hardware queues of their own: still serialised?
A="--steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --no-extra-blocks"
run() { echo -n "$1: "; env $2 python bench.py $A 2>&1 | grep '^{' | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['config']['step_launch_mode'][:20])"; }
run tape "YOLO_STEP_MODE=tape"
run graph "YOLO_STEP_MODE=graph"
run graph_q16 "YOLO_STEP_MODE=graph GPU_MAX_HW_QUEUES=16"
run eager "YOLO_STEP_MODE=eager"
run tape "YOLO_STEP_MODE=tape"
