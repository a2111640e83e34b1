#!/bin/bash
# is the forced-RCCL step slower because its streams collide on the hardware queues (GPU_MAX_HW_QUEUES, default 4)?
A="--steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --no-extra-blocks"
E="YOLO_DP_FORCE=1 MASTER_ADDR=127.0.0.1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0"
run() { echo -n "$1: "; env $2 python bench.py $A 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'])"; }
run plain "X=1"
run forced "$E MASTER_PORT=29581"
run forced_hwq8 "$E MASTER_PORT=29582 GPU_MAX_HW_QUEUES=8"
run forced_hwq16 "$E MASTER_PORT=29583 GPU_MAX_HW_QUEUES=16"
run plain_hwq8 "GPU_MAX_HW_QUEUES=8"
run plain_hwq2 "GPU_MAX_HW_QUEUES=2"
