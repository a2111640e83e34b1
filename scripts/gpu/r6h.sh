#!/bin/bash
# round 4: parity tests with the tightened census bounds + the fp32-CPU yardstick of the unconditioned companion; DP readiness
O=gpurun_out/r6h; mkdir -p $O; rm -f gpurun_out/parity_census.jsonl
bash scripts/dp_readiness.sh $O/dp_readiness.json > $O/dp.log 2>&1; echo "dp rc=$?"; tail -3 $O/dp.log | cut -c1-2500
timeout -k 10 500 python -m pytest tests/test_gpu_dp.py -x -q -s -k "costs_no_more" > $O/dp_test.log 2>&1; echo "dp test rc=$?"; grep -E "plain|passed|failed|Error" $O/dp_test.log | cut -c1-600
timeout -k 10 900 python -m pytest tests/test_gpu_model.py -x -q -s -k "test_model_parity" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log
grep -E "unconditioned oracle|passed|failed|Error|assert" $O/tests.log | cut -c1-420 | tail -20
