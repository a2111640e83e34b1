#!/bin/bash
O=gpurun_out/r6z; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_loss.py tests/test_gpu_kats.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
python scripts/loss_bench.py 2>/dev/null | tee $O/loss_new.log
git stash -q 2>/dev/null
