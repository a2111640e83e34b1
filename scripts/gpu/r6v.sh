#!/bin/bash
# round 4 final-state profile: smoke, bench line, kernel stats (two streams / one stream), HBM traffic, conv PMC
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash scripts/round_profile.sh r04_e > gpurun_out/r04_e_round_profile.log 2>&1; echo "round_profile rc=$?"
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r04_e_bench.json').read().strip().splitlines()[-1])
print(j['value'], j['ms_per_step'], j['roofline']['kernel'], j['roofline']['frac'], j['roofline']['traffic'])
print(json.dumps(j['summary']))
PY
