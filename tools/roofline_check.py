"""Recomputes the roofline fractions of a bench line from a rocprofv3 kernel trace of THE SAME RUN, cut to its timed regions:

    python tools/roofline_check.py <timed-regions stats CSV (tools/trace_regions.py --csv=...)> <bench line JSON of that run> > profiles/r6_roofline_check.md

Per kernel family: launches and time per step from the CSV alone; algorithmic bytes per step from the line's per-template work counts
(mi_dmrecon_stats::n_eval_by_kernel ...: 300 B per patch-view evaluation + 75 B per patch; + 28 B per filled pixel and the
compulsory image bytes for the bulk family as a whole); fraction of the 8 TB/s HBM roof = bytes / time / 8e12.  Then the line's own
figures next to the recomputed ones: the profile reproduces the line if the kernel time per step is below ms_per_step and the bulk
fraction agrees within 5 %."""
import csv
import json
import sys

HBM = 8000.0
FAM_OF_TEMPLATE = {"fast": "k_optimize FAST", "follow": "k_optimize follow-up (single attempt)", "seed": "k_optimize seeds",
                   "loop": "k_optimize general (attempts in a row)", "spec": "k_optimize_spec", "latency": "k_optimize latency layout",
                   "tail": "k_tail", "front": "k_front"}


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    line = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    regions, steps = int(rows[0]["TimedRegions"]), int(rows[0]["StepsPerRegion"])
    n_steps = regions * steps
    fam = {}
    for r in rows:
        f = fam.setdefault(r["Family"], [0, 0.0])
        f[0] += int(r["Calls"]); f[1] += float(r["TotalDurationNs"]) / 1e6
    roof = line["roofline"]
    tmpl = roof["per_kernel_template"]
    print("# Roofline check: the bench line recomputed from the kernel trace of the same run (timed regions only)\n")
    print("Run: `%s` under `rocprofv3 --kernel-trace`; %d timed regions x %d steps; value %.1f depth-maps/s, ms_per_step %.3f "
          "(under the profiler).\n" % (line["config"]["workload"][:40] + "...", regions, steps, line["value"], line["ms_per_step"]))
    print("| kernel family | launches / step | ms / step (trace) | n_eval / step (line) | passes executed / step | algorithmic GB / step | GB/s | frac of 8 TB/s |")
    print("|---|---|---|---|---|---|---|---|")
    bulk_ms = bulk_b = tail_ms = tail_b = 0.0
    for kind, family in FAM_OF_TEMPLATE.items():
        if family not in fam and kind not in tmpl:
            continue
        calls, ms = fam.get(family, [0, 0.0])
        t = tmpl.get(kind, {})
        b = t.get("algorithmic_bytes_per_step", 0.0)
        ms_step = ms / n_steps
        gbs = b / (ms_step / 1e3) / 1e9 if ms_step > 0 else 0.0
        print("| %s | %.2f | %.4f | %.3e | %.3e | %.3f | %.0f | %.3f |" % (family, calls / n_steps, ms_step, t.get("n_eval_per_step", 0), t.get("n_pass_executed_per_step", 0),
                                                                       b / 1e9, gbs, gbs / HBM))
        if kind in ("tail", "front"):
            tail_ms += ms_step; tail_b += b
        else:
            bulk_ms += ms_step; bulk_b += b
    # the bulk family as the line accounts it: + 28 B per pixel filled there + the compulsory image bytes
    pk = roof["per_kernel"]["k_optimize<1> (host-visible rounds)"]
    line_bulk_b = pk["algorithmic_bytes_per_launch"] * pk["launches"] / max(line["steps"] * len(line["repeats"]), 1)
    opt_ms = bulk_ms + tail_ms
    print("\n| | from the trace + work counts | the line |")
    print("|---|---|---|")
    print("| optimise kernels, ms per step | %.3f (bulk %.3f + tail %.3f) | %.3f = ms_per_step (wall clock of a step) |" % (opt_ms, bulk_ms, tail_ms, line["ms_per_step"]))
    rb = line_bulk_b / (bulk_ms / 1e3) / 1e9 / HBM if bulk_ms > 0 else 0.0
    print("| bulk_kernel_frac | %.4f (%.3f GB per step incl. 28 B per filled pixel and the compulsory bytes / %.3f ms) | %.4f |" % (rb, line_bulk_b / 1e9, bulk_ms, roof["bulk_kernel_frac"]))
    allb = roof["algorithmic_bytes_per_launch"] * roof["launches"] / max(line["steps"] * len(line["repeats"]), 1)
    print("| frac (all optimise kernels) | %.4f | %.4f |" % (allb / (opt_ms / 1e3) / 1e9 / HBM if opt_ms > 0 else 0.0, roof["frac"]))
    ok_time = opt_ms <= line["ms_per_step"]
    ok_frac = roof["bulk_kernel_frac"] and abs(rb / roof["bulk_kernel_frac"] - 1.0) <= 0.05
    print("\nOptimise-kernel time per step in the trace %s the step's wall clock; recomputed bulk fraction %s 5 %% of the line's (%.1f %%)."
          % ("is below" if ok_time else "EXCEEDS", "within" if ok_frac else "NOT within", 100.0 * (rb / roof["bulk_kernel_frac"] - 1.0) if roof["bulk_kernel_frac"] else 0.0))
    print("\nOther kernels of the timed regions (ms per step): " + ", ".join("%s %.4f" % (k, v[1] / n_steps) for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1]) if k not in FAM_OF_TEMPLATE.values()))


if __name__ == "__main__":
    main()
