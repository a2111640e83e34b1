#!/bin/bash
# side streams verified by a concurrency probe: C4 inside bench.py, forced RCCL with 4 and 8 hardware queues, plain
A="--steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --no-extra-blocks"
E="YOLO_DP_FORCE=1 MASTER_ADDR=127.0.0.1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0"
run() { echo -n "$1: "; env $2 python bench.py $A 2>&1 | grep '^{' | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'])"; }
run plain "X=1"
run forced_q8 "$E MASTER_PORT=29581"
run forced_q4 "$E MASTER_PORT=29582 GPU_MAX_HW_QUEUES=4"
run forced_q4_probe "$E MASTER_PORT=29583 GPU_MAX_HW_QUEUES=4 YOLO_STREAM_PROBE=1"
run plain_q4 "GPU_MAX_HW_QUEUES=4"
echo "bench with configs:"
python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-timer 2>gpurun_out/r6p_bench.log | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], json.dumps(j['summary'])[:330])"
grep -i "warn\|overlap" gpurun_out/r6p_bench.log | head -5
