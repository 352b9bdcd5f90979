"""Kernel concurrency report from a rocprofv3 --kernel-trace CSV of the multi-stream bench: share of the steady-state
window in which throughput-layout (bulk) / latency-layout (tail) optimise kernels and the small sweep kernels run."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    n = r["Kernel_Name"]
    k = "bulk" if "k_optimize<1>" in n else "tail" if "k_optimize<16>" in n else "other"
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k))
ev.sort()
t0, t1 = ev[0][0], max(e for _, e, _ in ev)
lo = t0 + (t1 - t0) * 0.4
pts = []
for s, e, k in ev:
    if e < lo:
        continue
    pts.append((max(s, lo), 1, k)); pts.append((e, -1, k))
pts.sort()
cnt, state, last = collections.Counter(), collections.Counter(), lo
for t, d, k in pts:
    key = "+".join(x for x in ("bulk", "tail", "other") if cnt[x] > 0) or "idle"
    state[key] += t - last; last = t
    cnt[k] += d
tot = sum(state.values())
print("window %.1f ms" % ((t1 - lo) / 1e6))
for k, v in state.most_common():
    print("  %-18s %5.1f%%" % (k, 100 * v / tot))
d = collections.defaultdict(list)
for s, e, k in ev:
    if s > lo:
        d[k].append(e - s)
for k, v in d.items():
    print("  %-5s n=%5d avg %8.1f us total %7.1f ms" % (k, len(v), sum(v) / len(v) / 1e3, sum(v) / 1e6))
