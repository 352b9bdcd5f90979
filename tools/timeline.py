"""GPU timeline of a multi-threaded bench run from a rocprofv3 kernel trace (CSV): how much of the wall time has a bulk
kernel running, how many kernels overlap, what the queues wait for.

    rocprofv3 --kernel-trace --output-format csv -d DIR -o bench -- python bench.py ...
    python tools/timeline.py DIR/**/bench_kernel_trace.csv
"""
import csv, sys, glob, collections
import numpy as np

path = sys.argv[1] if len(sys.argv) > 1 else None
files = glob.glob(path, recursive=True) if path and "*" in path else [path]
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
rows.sort()
t0 = rows[0][0]; t1 = max(r[1] for r in rows)
def kind(n):
    if "k_optimize" in n: return "bulk" if "Li1E" in n or "<1" in n else "opt16"
    if "k_tail" in n: return "tail"
    return "other"
# steady-state window: drop the first and last 15 %
lo = t0 + 0.15 * (t1 - t0); hi = t1 - 0.15 * (t1 - t0)
ev = []
for s, e, n, q in rows:
    s2, e2 = max(s, lo), min(e, hi)
    if e2 <= s2: continue
    k = kind(n)
    ev.append((s2, 1, k)); ev.append((e2, -1, k))
ev.sort()
cnt = collections.Counter(); last = lo
busy = collections.Counter(); conc = collections.Counter(); bulk_conc = collections.Counter()
for t, d, k in ev:
    dt = t - last
    if dt > 0:
        tot = sum(cnt.values())
        conc[min(tot, 8)] += dt
        bulk_conc[min(cnt["bulk"], 4)] += dt
        for kk, v in cnt.items():
            if v > 0: busy[kk] += dt
        if cnt["bulk"] == 0 and cnt["tail"] > 0: busy["tail_only"] += dt
    cnt[k] += d; last = t
W = hi - lo
print("window %.1f ms, %d kernel launches in the trace" % (W / 1e6, len(rows)))
print("time with >= 1 kernel of a kind running:", {k: "%.1f %%" % (100 * v / W) for k, v in busy.items()})
print("kernels running at once:", {k: "%.1f %%" % (100 * v / W) for k, v in sorted(conc.items())})
print("bulk kernels running at once:", {k: "%.1f %%" % (100 * v / W) for k, v in sorted(bulk_conc.items())})
dur = collections.defaultdict(list)
for s, e, n, q in rows:
    if s >= lo and e <= hi: dur[kind(n)].append((e - s) / 1e3)
for k, v in dur.items():
    v = np.array(v); print("%-6s n %6d  mean %8.1f us  median %8.1f  p90 %8.1f  sum %8.1f ms" % (k, len(v), v.mean(), np.median(v), np.percentile(v, 90), v.sum() / 1e3))
names = collections.Counter()
for s, e, n, q in rows:
    if s >= lo and e <= hi: names[n[:60]] += (e - s)
for n, v in names.most_common(8): print("  %6.1f ms  %s" % (v / 1e6, n))
