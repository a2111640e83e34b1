#!/bin/bash
python bench.py --steps 10 --warmup 3 > gpurun_out/r03_d_bench.json 2> gpurun_out/r03_d_bench.log; python scripts/bench_line.py gpurun_out/r03_d_bench.json
