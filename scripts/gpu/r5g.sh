#!/bin/bash
python bench.py --steps 10 --warmup 3 > gpurun_out/r03_e_bench.json 2> gpurun_out/r03_e_bench.log; python scripts/bench_line.py gpurun_out/r03_e_bench.json
python scripts/bench_configs.py > gpurun_out/r03_e_configs.jsonl 2>gpurun_out/r03_e_configs.err; cut -c1-200 gpurun_out/r03_e_configs.jsonl | head -5
for V in 1 0 1 0; do echo -n "tight=$V: "; YOLO_BN_TIGHT_BOUND=$V python scripts/bench_configs.py c4 2>&1 | grep images_per_s | cut -c1-170; done
