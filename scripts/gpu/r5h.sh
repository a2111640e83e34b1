#!/bin/bash
R=$PWD; O=$R/gpurun_out/r5h; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 400 python scripts/pair_matrix.py $O/pair_matrix.json > $O/pair.log 2>&1; echo "pair rc $?"; tail -5 $O/pair.log
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5h/pair_matrix.json"))
print("priority range", d["stream_priority_range"])
for L,v in d["layers"].items():
    print(L, "wgrad alone", v["filter_gradient_alone_us"])
    for k,r in v["beside"].items():
        print("   %-28s alone %7.1f us | " % (k, r["alone_us"]) + " | ".join("%s: %.3f" % (t, r[t]["ratio"]) for t in r if isinstance(r[t], dict)))
PY
