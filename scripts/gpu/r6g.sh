#!/bin/bash
# round 4: parity hardening (census, unconditioned companion, element-wise planes test) + the new bench blocks
O=gpurun_out/r6g; mkdir -p $O; rm -f gpurun_out/parity_census.jsonl
timeout -k 10 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_conv.py -x -q -s -k "test_model_parity or small_magnitude" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log
grep -E "per-channel|conv1 output|BN \+ Leaky|conv2 output|unconditioned oracle|passed|failed|Error|assert" $O/tests.log | cut -c1-400 | tail -40
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.log; echo "bench rc=$?"
tail -12 $O/bench.log
python -c "
import json; j=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(json.dumps(j['summary'])); print(json.dumps(j['strict_fp32'])[:600]); print(json.dumps(j['forward']['darknet53'])); print(json.dumps(j['configs'])[:1800])"
