#!/bin/bash
O=gpurun_out/r7a; mkdir -p $O
timeout -k 10 1150 python -m pytest tests -m gpu -x -q --durations=6 > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log
tail -12 $O/tests.log
