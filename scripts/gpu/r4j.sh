#!/bin/bash
bash scripts/pmc_round.sh r03_c > gpurun_out/r03_c_pmc_round.log 2>&1; echo "pmc_round rc $?"; tail -3 gpurun_out/r03_c_pmc_round.log
python - <<'PY'
import json
j=json.load(open('gpurun_out/r03_c_conv_pmc.json'))
for k,v in (j.items() if isinstance(j,dict) else enumerate(j)):
    print(k, json.dumps(v)[:300])
PY
