#!/bin/bash
R=$PWD; O=$R/gpurun_out/r5e; mkdir -p $O; export TMPDIR=/tmp
rm -f gpurun_out/parity_ratios.jsonl gpurun_out/parity_census.jsonl
nproc > $O/nproc.txt
python -m pytest tests -q -m gpu --durations=25 > $O/tests.log 2>&1; echo "gpu suite rc $?"; tail -45 $O/tests.log
