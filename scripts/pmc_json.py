"""Merge the per-pass lines of scripts/pmc_round.sh into one JSON with derived figures per kernel/layer:
MFMA busy % (SQ_VALU_MFMA_BUSY_CYCLES per SIMD / GRBM_GUI_ACTIVE per XCD), effective clock, LDS active %,
L2 hit rate, beyond-L2 bytes (gfx950: FETCH_SIZE counts 64 B per 128-B request -> x2; KiB units)."""
import json, sys
rows = {}
for line in open(sys.argv[1]):
    j = json.loads(line)
    e = rows.setdefault(j["label"], {"label": j["label"], "mode": j["mode"], "layer_H_Cin_Cout_k_s_N": j["layer_H_Cin_Cout_k_s_N"],
                                     "kernel": j["pass"]["kernel"], "counters_per_launch": {}, "durations_us": []})
    e["counters_per_launch"].update(j["pass"]["counters_per_launch"])
    if j["pass"]["avg_duration_us_under_profiler"]:
        e["durations_us"].append(j["pass"]["avg_duration_us_under_profiler"])
    if j["pass"]["kernel"]:
        e["kernel"] = j["pass"]["kernel"]
out = []
for e in rows.values():
    c = e["counters_per_launch"]
    dur = sorted(e["durations_us"])[len(e["durations_us"]) // 2] if e["durations_us"] else None
    h, cin, cout, k, s, n = map(int, e["layer_H_Cin_Cout_k_s_N"].split(","))
    ho = h // s
    flops = 2.0 * n * ho * ho * cout * k * k * cin
    d = {"avg_duration_us_under_profiler": dur, "algorithmic_gflop": round(flops / 1e9, 2)}
    if dur:
        d["algorithmic_tflops_under_profiler"] = round(flops / dur / 1e6, 1)
        d["frac_of_833_under_profiler"] = round(flops / dur / 1e6 / (2500 / 3), 4)
    if "GRBM_GUI_ACTIVE" in c:
        cyc = c["GRBM_GUI_ACTIVE"] / 8
        d["gpu_cycles"] = round(cyc)
        if dur:
            d["effective_clock_ghz"] = round(cyc / dur / 1e3, 3)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            d["mfma_busy_pct"] = round(100 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc, 1)
        if "SQ_LDS_IDX_ACTIVE" in c:
            d["lds_active_pct"] = round(100 * c["SQ_LDS_IDX_ACTIVE"] / 256 / cyc, 1)   # per-CU LDS cycles (quad-cycle units excluded)
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
        d["l2_hit_rate"] = round(c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1), 4)
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        d["beyond_l2_bytes"] = int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)
    if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_pct_of_lds_cycles"] = round(100 * c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 1)
    e["derived"] = d
    del e["durations_us"]
    out.append(e)
print(json.dumps({"how": "scripts/pmc_round.sh: rocprofv3 --pmc <group> --kernel-trace -- scripts/hip_probe/conv_bench.bin ... "
                         "(one pass per counter group, 12 launches each, random normal operands); profiled passes run at a "
                         "lower clock than un-profiled ones (MI355X_MICROARCH.md, DVFS give-back item 2)",
                  "kernels": out}, indent=1))
