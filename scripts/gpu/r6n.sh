#!/bin/bash
# round 4 profile of the current commit: bench line, kernel stats (two streams / one stream), HBM traffic passes, conv PMC passes
bash scripts/round_profile.sh r04_c > gpurun_out/r04_c_round_profile.log 2>&1; echo "round_profile rc=$?"
tail -c 600 gpurun_out/r04_c_round_profile.log
bash scripts/pmc_round.sh r04_c > gpurun_out/r04_c_pmc_round.log 2>&1; echo "pmc_round rc=$?"
python - <<'PY'
import json
j=json.load(open('gpurun_out/r04_c_conv_pmc.json'))
for k in j['kernels']:
    d=k['derived']; print(k['label'], k['mode'], d.get('avg_duration_us_under_profiler'), d.get('frac_of_833_under_profiler'), d.get('mfma_busy_pct'), d.get('effective_clock_ghz'), d.get('l2_hit_rate'), d.get('lds_active_pct'), d.get('lds_bank_conflict_pct_of_lds_cycles'), d.get('beyond_l2_bytes'))
PY
