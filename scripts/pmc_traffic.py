"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; collected separately, as
MI355X_MICROARCH.md prescribes: they do not fit one pass) into per-kernel HBM bytes per launch.
gfx950 correction: FETCH_SIZE tallies 128-B requests at 64 B for wide coalesced reads -> x2; both
counters are in KiB. usage: pmc_traffic.py <fetch_dir> <write_dir> <out.json>"""
import collections, csv, glob, json, re, sys


def agg(d, counter):
    f = glob.glob(f"{d}/*/*counter_collection.csv")[0]
    a = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            a[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return a


def short(name):
    m = re.match(r"(?:void )?yolo::(\w+)(<[^>]*>)?", name)
    if not m:
        return name
    s = (m.group(1) + (m.group(2) or "")).replace(" ", "")
    # the diagnostic knock-out parameter (0 in production) is not part of the variant name bench.py uses
    return re.sub(r"^(gather_conv_planes_kernel<\d+,\d+,\d+,\d+),0>$", r"\1>", s)


fe, wr = agg(sys.argv[1], "FETCH_SIZE"), agg(sys.argv[2], "WRITE_SIZE")
out = {}
for k, v in fe.items():
    w = wr.get(k, [])
    f_avg = sum(v) / len(v)
    w_avg = sum(w) / len(w) if w else 0.0
    out[short(k)] = {"launches": len(v), "fetch_size_kib_avg": round(f_avg, 1), "write_size_kib_avg": round(w_avg, 1),
                     "hbm_bytes_per_launch": int((2 * f_avg + w_avg) * 1024)}
json.dump({"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --steps 2 --warmup 1 "
                      "--no-cpu-baseline --no-kernel-timer (two separate passes)",
           "correction": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE reads 1/2 of wide coalesced reads; "
                         "cross-check: bn_act_fwd_kernel reads x (+ residual in 23/72 launches) = 1.32 x its writes -> expected "
                         "90 MB, corrected counter 94 MB)",
           "kernels": out}, open(sys.argv[3], "w"), indent=1)
print("wrote", sys.argv[3], len(out), "kernels")
