#!/bin/bash
# where do the 4 ms of the forced-RCCL step go? plain / forced / forced without the collective call / one bucket / many buckets
O=gpurun_out/r6i; mkdir -p $O
A="--steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --no-extra-blocks"
E="YOLO_DP_FORCE=1 MASTER_ADDR=127.0.0.1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0"
run() { echo -n "$1: "; env $2 python bench.py $A 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], (j.get('dp_trace') or {}).get('buckets'))"; }
run plain "X=1"
run forced "$E MASTER_PORT=29581"
run forced_dry "$E MASTER_PORT=29582 YOLO_DP_DRYRUN=1"
run forced_1bucket "$E MASTER_PORT=29583 YOLO_DP_BUCKET_MB=4096"
run forced_8MB "$E MASTER_PORT=29584 YOLO_DP_BUCKET_MB=8"
run forced_eager "$E MASTER_PORT=29585 YOLO_STEP_MODE=eager"
run plain_eager "YOLO_STEP_MODE=eager"
run plain "X=1"
