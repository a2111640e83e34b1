#!/bin/bash
# round-3 final profiles: bench line, rocprofv3 kernel stats (two streams / one stream), PMC traffic, other configurations
bash scripts/round_profile.sh r03_d > gpurun_out/r03_d_round_profile.log 2>&1; echo "round_profile rc $?"
python scripts/bench_configs.py > gpurun_out/r03_d_configs.jsonl 2>gpurun_out/r03_d_configs.err; cat gpurun_out/r03_d_configs.jsonl | cut -c1-220
python scripts/bench_line.py gpurun_out/r03_d_bench.json
python scripts/kstats_summary.py gpurun_out/r03_d_bench_kernel_stats.csv 14
python scripts/kstats_summary.py gpurun_out/r03_d_serial_kernel_stats.csv 8
