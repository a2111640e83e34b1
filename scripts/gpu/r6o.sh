#!/bin/bash
# C4 jumped from 39.8 to 58.8 ms inside bench.py's configs block: hardware queues? stream creation order?
for Q in 4 8; do echo "GPU_MAX_HW_QUEUES=$Q alone:"; GPU_MAX_HW_QUEUES=$Q python scripts/bench_configs.py c4 2>/dev/null | cut -c1-200; done
echo "inside bench.py (C3 first), queues 8 then 4:"
for Q in 8 4; do GPU_MAX_HW_QUEUES=$Q python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-timer 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], json.dumps(j['summary'])[:400])"; done
