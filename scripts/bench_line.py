"""one-line summary of a bench.py output file"""
import json, sys
for f in sys.argv[1:]:
    d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    fw = d.get("forward") or {}
    r = d.get("roofline") or {}
    print(f, d["value"], "img/s", d["ms_per_step"], "ms; fwd", fw.get("ms"), "conv frac", fw.get("conv_frac_of_peak"), "| dominant", r.get("kernel"), r.get("frac"))
