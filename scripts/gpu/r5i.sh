#!/bin/bash
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | grep -v amdgpu | tail -2 | cut -c1-400
python bench.py 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(j['metric'], j['value'], j['n_gpus'], j['steps'], j['warmup'], j['roofline']['frac'], j['cpu_baseline']['value'])"
