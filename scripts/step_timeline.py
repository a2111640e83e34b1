"""Timeline of ONE training step from a rocprofv3 --kernel-trace csv: per hardware queue busy time in the forward and
backward phases, when each queue finishes, the idle gaps of the compute queue, and kernel time per family on each queue.
Step boundaries = the Adam kernel; forward / backward boundary = the first loss kernel.
usage: step_timeline.py <dir or csv> [step_from_end=1] [--json out.json]"""
import collections, csv, glob, json, os, sys
src = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else 1
out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
files = [src] if os.path.isfile(src) else glob.glob(src + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]))
rows.sort()
adam = [i for i, r in enumerate(rows) if "adam" in r[2]]
assert len(adam) >= back + 1, f"only {len(adam)} optimizer launches in the trace"
lo, hi = adam[-back - 1] + 1, adam[-back] + 1
step = rows[lo:hi]
t0 = step[0][0]
t_end = max(r[1] for r in step)
loss_i = next(i for i, r in enumerate(step) if "loss" in r[2])
t_loss = step[loss_i][0]


def fam(k):
    k = k.replace("void yolo::", "")
    for key, name in (("wgrad_win_reduce", "wgrad reduce"), ("wgrad_reduce", "wgrad reduce"), ("colsum", "wgrad reduce"),
                      ("wgrad_win", "wgrad x-window"), ("wgrad", "wgrad per-tap"),
                      ("conv_win_kernel", "conv window"), ("gather_conv", "conv per-tap/1x1"), ("conv_split_reduce", "conv split reduce"),
                      ("bn_bwd_reduce", "bn bwd reduce"), ("bn_bwd_sum", "bn bwd sum"), ("bn_bwd_apply", "bn bwd apply"),
                      ("bn_finalize", "bn finalize"), ("bn_act_fwd", "bn fwd apply"), ("stem", "stem"), ("loss", "loss"),
                      ("adam", "adam"), ("split_planes", "split planes"), ("transpose", "filter prep"), ("head_act", "head act")):
        if key in k:
            return name
    return "other"


queues = collections.OrderedDict()
for s, e, k, q in step:
    queues.setdefault(q, []).append((s, e, k))
main_q = max(queues, key=lambda q: len(queues[q]))
print(f"step: {len(step)} launches, wall {(t_end - t0) / 1e6:.3f} ms; forward {(t_loss - t0) / 1e6:.3f} ms, backward+opt {(t_end - t_loss) / 1e6:.3f} ms")
res = {"launches": len(step), "wall_ms": (t_end - t0) / 1e6, "forward_ms": (t_loss - t0) / 1e6, "queues": {}}
for q, ks in queues.items():
    busy_f = sum(min(e, t_loss) - s for s, e, k in ks if s < t_loss)
    busy_b = sum(e - max(s, t_loss) for s, e, k in ks if e > t_loss)
    first, last = ks[0][0], max(e for s, e, k in ks)
    fams = collections.Counter()
    for s, e, k in ks:
        fams[fam(k)] += e - s
    tag = "compute" if q == main_q else "side"
    print(f"queue {q} ({tag}): {len(ks)} launches, busy fwd {busy_f / 1e6:.3f} ms, busy bwd {busy_b / 1e6:.3f} ms, "
          f"first +{(first - t0) / 1e6:.3f} ms, last end +{(last - t0) / 1e6:.3f} ms")
    print("    " + ", ".join(f"{n} {v / 1e6:.2f}" for n, v in fams.most_common()))
    res["queues"][q] = {"role": tag, "launches": len(ks), "busy_fwd_ms": busy_f / 1e6, "busy_bwd_ms": busy_b / 1e6,
                        "last_end_ms": (last - t0) / 1e6, "families_ms": {n: v / 1e6 for n, v in fams.items()}}
# idle gaps of the compute queue
ks = queues[main_q]
gaps = []
for (s0, e0, k0), (s1, e1, k1) in zip(ks, ks[1:]):
    if s1 > e0:
        gaps.append((s1 - e0, k0, k1, (e0 - t0) / 1e6))
tot_f = sum(g for g, k0, k1, t in gaps if t * 1e6 + t0 < t_loss)
tot_b = sum(g for g, k0, k1, t in gaps if t * 1e6 + t0 >= t_loss)
print(f"compute queue idle between its own launches: forward {tot_f / 1e6:.3f} ms, backward {tot_b / 1e6:.3f} ms ({len(gaps)} gaps)")
hist = collections.Counter(min(int(g / 1e3) // 2 * 2, 40) for g, *_ in gaps)
print("    gap histogram (us bucket: count):", sorted(hist.items()))
gaps.sort(reverse=True)
for g, k0, k1, t in gaps[:14]:
    print(f"    {g / 1e3:7.1f} us at +{t:.3f} ms  {fam(k0)} -> {fam(k1)}")
res["compute_idle_fwd_ms"], res["compute_idle_bwd_ms"] = tot_f / 1e6, tot_b / 1e6
# union busy over all queues
iv = sorted((s, e) for s, e, k, q in step)
busy, cs, ce = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > ce:
        busy += ce - cs
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print(f"GPU busy (union of all queues) {busy / 1e6:.3f} ms of {(t_end - t0) / 1e6:.3f}; kernel-time sum {sum(e - s for s, e in iv) / 1e6:.3f} ms")
res["union_busy_ms"], res["kernel_sum_ms"] = busy / 1e6, sum(e - s for s, e in iv) / 1e6
if out_json:
    json.dump(res, open(out_json, "w"), indent=1)
