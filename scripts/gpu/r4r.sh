#!/bin/bash
R=$PWD; O=$R/gpurun_out/r4r; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-timer > $O/prof.log 2>&1 || echo "prof failed"
cd $R
python scripts/trace_gaps.py $O/kt
YOLO_BWD_OVERLAP=0 bash -c "cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt1 -- python3 $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-timer > $O/prof1.log 2>&1" || echo "prof1 failed"
python scripts/trace_gaps.py $O/kt1
rm -rf $O/kt $O/kt1
