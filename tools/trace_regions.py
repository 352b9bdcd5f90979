"""Cuts a rocprofv3 kernel trace (`--kernel-trace --output-format csv`: *_kernel_trace.csv) of a bench.py run to its TIMED
REGIONS -- bench.py brackets every region with a `k_region_mark` dispatch (mi_dmrecon_debug_region_mark) -- and says, from the
dispatches' own start / end timestamps: per kernel family the launches, the summed and the average duration; the time some
kernel was running (union of the intervals), the time NOTHING was running (gaps: launch latency, host round trips, drains
are inside the intervals), and how many kernels ran side by side (time-weighted).

usage: python tools/trace_regions.py <kernel_trace.csv> [steps per region] [regions to keep, default all] [> summary.json]
The warm-up calls and whatever bench.py runs after the timed regions (one_call, variants) lie outside the marks and are dropped:
what is left is exactly what `value` is measured over."""
import collections
import csv
import json
import sys


def short(name):
    n = name.replace("void ", "")
    i = n.find("(")
    return n if i < 0 else n[:i]


def family(k):
    if "k_front<" in k:
        return "k_front"
    if "k_tail<" in k:
        return "k_tail"
    if "k_optimize_spec<" in k:
        return "k_optimize_spec"
    if "k_optimize<" in k:
        if "Lay<16, 4>" in k or "Lay<8, 8>" in k or "Lay<16,4>" in k or "Lay<8,8>" in k:
            return "k_optimize latency layout"
        inside = k[k.find("k_optimize<"):]
        # template arguments after the layout: FAST, SINGLE, SEED
        args = inside[inside.rfind(">,") + 2:].strip("> ").replace(" ", "").split(",") if ">," in inside else []
        flags = [a in ("true", "1", "(bool)1") for a in args]
        if len(flags) >= 3 and flags[2]:
            return "k_optimize seeds"
        if flags and flags[0]:
            return "k_optimize FAST"
        if len(flags) >= 2 and flags[1]:
            return "k_optimize follow-up (single attempt)"
        return "k_optimize general (attempts in a row)"
    for s in ("k_generate", "k_apply_spec", "k_apply_seeds", "k_apply", "k_round_report", "k_flatten", "k_emit_changed", "k_front_split",
              "k_front_commit", "k_unpack_jobs", "k_gvs", "k_region_mark"):
        if s in k:
            return s
    return "other"


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--csv=")]
    csv_out = next((a[6:] for a in sys.argv[1:] if a.startswith("--csv=")), None)   # --csv=FILE: per-kernel stats of the timed regions
    path = args[0]
    steps = int(args[1]) if len(args) > 1 else 20
    keep = int(args[2]) if len(args) > 2 else 0
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", ""), r.get("Stream_Id", "")))
    rows.sort()
    marks = [r for r in rows if "k_region_mark" in r[2]]
    regions = [(marks[i][1], marks[i + 1][0]) for i in range(0, len(marks) - 1, 2)]
    if keep:
        regions = regions[:keep]
    out = {"trace": path, "steps_per_region": steps, "regions": []}
    tot = collections.defaultdict(lambda: [0, 0.0, 0.0])
    by_name = collections.defaultdict(lambda: [0, 0, 1 << 62, 0])          # kernel name -> calls, total ns, min, max (timed regions only)
    for (t0, t1) in regions:
        inside = [r for r in rows if r[0] >= t0 and r[1] <= t1]
        fam = collections.defaultdict(lambda: [0, 0.0, 0.0])
        ev = []
        for (a, b, k, q, s) in inside:
            bn = by_name[k]
            bn[0] += 1; bn[1] += b - a; bn[2] = min(bn[2], b - a); bn[3] = max(bn[3], b - a)
            f = fam[family(k)]
            f[0] += 1; f[1] += (b - a) / 1e6; f[2] = max(f[2], (b - a) / 1e6)
            ev.append((a, 1)); ev.append((b, -1))
        ev.sort()
        depth, last, busy, conc = 0, t0, 0.0, collections.defaultdict(float)
        for (t, d) in ev:
            conc[depth] += (t - last) / 1e6
            if depth > 0:
                busy += (t - last) / 1e6
            depth += d; last = t
        conc[0] += (t1 - last) / 1e6
        wall = (t1 - t0) / 1e6
        # the patch-optimisation kernels alone: union and sum (what ms_bulk_kernel / ms_front_kernel add up from)
        opt = [(a, b) for (a, b, k, q, s) in inside if family(k).startswith("k_optimize") or family(k) in ("k_tail", "k_front")]
        opt_sum = sum(b - a for a, b in opt) / 1e6
        reg = {"wall_ms": wall, "ms_per_step": wall / steps, "some_kernel_running_ms": busy, "no_kernel_running_ms": wall - busy,
               "kernels_side_by_side_ms": {str(k): round(v, 3) for k, v in sorted(conc.items())},
               "optimise_kernels_sum_ms": opt_sum, "optimise_kernels_sum_ms_per_step": opt_sum / steps,
               "queues": len(set(q for (_, _, _, q, _) in inside)),
               "families": {k: {"launches": v[0], "sum_ms": round(v[1], 3), "avg_ms": round(v[1] / max(v[0], 1), 4), "max_ms": round(v[2], 3),
                                "sum_ms_per_step": round(v[1] / steps, 4)} for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1])}}
        out["regions"].append(reg)
        for k, v in fam.items():
            tot[k][0] += v[0]; tot[k][1] += v[1]; tot[k][2] = max(tot[k][2], v[2])
    n = max(len(regions), 1)
    out["all_regions"] = {"regions": len(regions), "steps": steps * len(regions),
                          "wall_ms_per_step": sum(r["wall_ms"] for r in out["regions"]) / (steps * n),
                          "optimise_kernels_sum_ms_per_step": sum(r["optimise_kernels_sum_ms"] for r in out["regions"]) / (steps * n),
                          "no_kernel_running_ms_per_step": sum(r["no_kernel_running_ms"] for r in out["regions"]) / (steps * n),
                          "families_ms_per_step": {k: round(v[1] / (steps * n), 4) for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1])},
                          "families_launches_per_region": {k: v[0] / n for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1])}}
    if csv_out:
        # the layout of rocprofv3's own kernel_stats.csv, over the dispatches BETWEEN the region marks only
        total = sum(v[1] for v in by_name.values()) or 1
        with open(csv_out, "w") as f:
            f.write('"Name","Family","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","TimedRegions","StepsPerRegion"\n')
            for k, v in sorted(by_name.items(), key=lambda kv: -kv[1][1]):
                f.write('"%s","%s",%d,%d,%.1f,%.2f,%d,%d,%d,%d\n' % (k, family(k), v[0], v[1], v[1] / max(v[0], 1), 100.0 * v[1] / total, v[2], v[3], len(regions), steps))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
