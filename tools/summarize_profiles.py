"""Turns gpurun_out/<tag>/ (written by tools/collect_profiles.sh on the GPU box) into the committed summaries
under profiles/: kernel stats CSVs (rocprofv3 --kernel-trace --stats), bench JSON lines, a PMC table and the
HBM-side traffic per step that bench.py reports as `roofline.traffic`.

Counter units (profiles/r2_pmc_calibration.md, tools/pmc_calib.sh): FETCH_SIZE is reported in KB of 64 B per
fabric read request, but every request moves 128 B -- for coalesced streams AND for scattered 16-byte gathers (one
request per record) -- so read bytes = FETCH_SIZE x 1024 x 2.  WRITE_SIZE is exact as reported (x 1024)."""
import csv
import collections
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r6"
src = os.path.join("gpurun_out", tag)
dst = "profiles"
FETCH_CORR = 2.0
# steps (passes over the 20 views) behind each PMC run of tools/collect_profiles.sh, warm-up calls included
STEPS = {"pmc": 3,      # --steps 2 --warmup 1 --streams 1 --steps-per-call 1 --no-one-call: three calls of one step
         "pmcd": 40}    # --steps 20 --warmup 1 --no-one-call: the DRIVER's plan, 4 host threads x 5 steps per call -- one warm-up
                        # batch and one timed batch of 400 views each
# the call plans the PMC passes were taken at, named exactly as bench.py names the plan of a run (`this_run_plan`)
PLAN = {"pmc": "1 host thread(s), 1 step(s) per call", "pmcd": "4 host thread(s), 5 step(s) per call"}
for n in ("1thread", "default", "driver", "1thread_5steps"):
    f = os.path.join(src, "bench_%s.json" % n)
    if os.path.exists(f) and os.path.getsize(f):
        shutil.copy(f, os.path.join(dst, "%s_bench_%s.json" % (tag, n)))
for name, dstname in (("lone_calls.json", "lone_calls.json"), ("valu_rate.txt", "valu_rate.txt"), ("round_trace_c3.txt", "round_trace_c3.txt"),
                      ("app_c3_timing.txt", "app_c3_timing.txt"), ("cold_call.txt", "cold_call.txt"), ("big_batch.txt", "big_batch.txt"),
                      ("strong_2ranks_one_gpu.json", "strong_2ranks_one_gpu.json"), ("bench_c2.json", "bench_c2.json"), ("bench_c5.json", "bench_c5.json")):
    f = os.path.join(src, name)
    if os.path.exists(f) and os.path.getsize(f):
        shutil.copy(f, os.path.join(dst, "%s_%s" % (tag, dstname)))
for s in ("s1", "s3"):
    # kernel statistics of the TIMED REGIONS only (tools/trace_regions.py --csv), their per-family summary, the bench line of that
    # profiled run and the roofline fractions recomputed from the two (tools/roofline_check.py)
    plan = "1thread_1step" if s == "s1" else "4threads_5steps"
    for name, dstname in (("%s_timed_regions_stats.csv" % s, "%s_kernel_stats_timed_regions_%s.csv" % (tag, plan)),
                          ("%s_timed_regions.json" % s, "%s_kernel_regions_%s.json" % (tag, plan)),
                          ("%s_line.json" % s, "%s_bench_profiled_%s.json" % (tag, plan)),
                          ("%s_roofline_check.md" % s, "%s_roofline_check%s.md" % (tag, "" if s == "s3" else "_1thread"))):
        f = os.path.join(src, name)
        if os.path.exists(f) and os.path.getsize(f):
            shutil.copy(f, os.path.join(dst, dstname))


def short(name):
    return name.split("(")[0].replace("void ", "")


def family(k):
    if "k_tail<" in k or "k_front<" in k:
        return "k_tail + k_front (tail rounds)"
    if "k_optimize<" in k or "k_optimize_spec<" in k:
        return "k_optimize<1> (host-visible rounds)"
    return None


def collect(prefix):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    regs = {}
    for d in sorted(glob.glob(os.path.join(src, prefix + "_*"))):
        f = os.path.join(d, "bench_counter_collection.csv")
        if not os.path.isdir(d) or not os.path.exists(f):
            continue
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            a = acc[k][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"]); a[1] += 1
            regs[k] = (row["VGPR_Count"], row["Accum_VGPR_Count"], row["SGPR_Count"], row["LDS_Block_Size"], row["Scratch_Size"])
    return acc, regs


HEAD = ["| kernel | launches | FETCH MB | WRITE MB | L2 hit % | VALU-active % of wave cycles | wave-cycles/VALU inst | wait-any % | LDS-active % | LDS bank-conflict % of LDS-active | L1 accesses | L1->L2 read req | L1 pending-stall % of L1 busy | VGPR/AGPR/SGPR | LDS B | scratch B |",
        "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]


def table(acc, regs):
    out = []
    for k in sorted(acc, key=lambda k: -acc[k].get("FETCH_SIZE", [0, 0])[0]):
        c = acc[k]
        if k.startswith("__amd"):
            continue
        def avg(n):
            return c[n][0] / c[n][1] if n in c and c[n][1] else None
        n = max(v[1] for v in c.values())
        hit, miss = avg("TCC_HIT_sum"), avg("TCC_MISS_sum")
        act = avg("SQ_ACTIVE_INST_VALU")
        wc, valu, wany = avg("SQ_WAVE_CYCLES"), avg("SQ_INSTS_VALU"), avg("SQ_WAIT_ANY")
        lds_a, lds_c = avg("SQ_ACTIVE_INST_LDS"), avg("SQ_LDS_BANK_CONFLICT")
        f = lambda v, s=1.0, fmt="%.3f": "-" if v is None else fmt % (v * s)
        l1a, l1r, l1p, l1g = avg("TCP_TOTAL_CACHE_ACCESSES_sum"), avg("TCP_TCC_READ_REQ_sum"), avg("TCP_PENDING_STALL_CYCLES_sum"), avg("TCP_GATE_EN1_sum")
        out.append("| `%s` | %d | %s | %s | %s | %s | %s | %s | %s | %s | %s | %s | %s | %s | %s | %s |" % (
            k, n, f(avg("FETCH_SIZE"), FETCH_CORR / 1024), f(avg("WRITE_SIZE"), 1 / 1024),
            "-" if hit is None or hit + miss == 0 else "%.1f" % (100 * hit / (hit + miss)),
            "-" if not wc or act is None else "%.1f" % (100 * act / wc),
            "-" if not valu or wc is None else "%.2f" % (wc / valu),
            "-" if not wc or wany is None else "%.1f" % (100 * wany / wc),
            "-" if not wc or lds_a is None else "%.1f" % (100 * lds_a / wc),
            "-" if not lds_a or lds_c is None else "%.1f" % (100 * lds_c / lds_a),
            "-" if l1a is None else "%.3g" % l1a, "-" if l1r is None else "%.3g" % l1r,
            "-" if not l1g or l1p is None else "%.1f" % (100 * l1p / l1g),
            "/".join(regs[k][:3]), regs[k][3], regs[k][4]))
    return out


acc, regs = collect("pmc")
lines = ["# %s -- PMC counters of the C3 bench (rocprofv3 --pmc, one counter group per run; tools/collect_profiles.sh)" % tag, "",
         "FETCH MB = FETCH_SIZE x 1024 x 2 (every fabric read request moves 128 B, the counter tallies",
         "64 B: calibrated on known byte counts in this kernel's own access pattern, profiles/r2_pmc_calibration.md);",
         "WRITE MB = WRITE_SIZE x 1024 (exact).  Per launch averages.", "",
         "## One lone 20-view call: `python bench.py --steps 2 --warmup 1 --streams 1 --steps-per-call 1` (3 reconstructions of the scene)", ""] + HEAD + table(acc, regs)
accd, regsd = collect("pmcd")
if accd:
    lines += ["", "## The driver's plan: `python bench.py --steps 20 --warmup 1` (4 host threads x 5 steps per call: two 400-view batches)", ""] + HEAD + table(accd, regsd)

# HBM-side bytes per STEP of the two optimise kernel families, at both call plans (bench.py divides by ITS launch
# counts: rocprofv3 also sees the empty tail rounds that the host enqueues blind)
traffic = {"correction": "read bytes = FETCH_SIZE x 1024 x 2 (128-byte requests tallied at 64 B; calibrated, profiles/r2_pmc_calibration.md); "
                         "written bytes = WRITE_SIZE x 1024", "plans": {}}
for prefix, plan in (("pmc", PLAN["pmc"]), ("pmcd", PLAN["pmcd"])):
    a, _ = (acc, regs) if prefix == "pmc" else collect("pmcd")
    fam = collections.defaultdict(lambda: [0.0, 0.0])
    for k in a:
        fm = family(k)
        if fm and "FETCH_SIZE" in a[k] and "WRITE_SIZE" in a[k]:
            fam[fm][0] += a[k]["FETCH_SIZE"][0] * 1024.0 * FETCH_CORR
            fam[fm][1] += a[k]["WRITE_SIZE"][0] * 1024.0
    if fam:
        traffic["plans"][plan] = {fm: {"read_bytes_per_step": v[0] / STEPS[prefix], "written_bytes_per_step": v[1] / STEPS[prefix]} for fm, v in fam.items()}
        lines += ["", "HBM-side traffic per step (20 depth maps), %s:" % plan] + [
            "  %s: %.1f MB read + %.1f MB written" % (fm, v[0] / STEPS[prefix] / 1e6, v[1] / STEPS[prefix] / 1e6) for fm, v in sorted(fam.items())]
# executed VALU wave-instructions of the bulk kernels per wavefront pass (64 patch-view passes): SQ_INSTS_VALU of a PMC pass
# over the passes its bench line counted on the device, at both call plans -- bench.py's `valu_issue` roof reads them here
def valu_per_wave_pass(prefix, steps_all):
    log = os.path.join(src, "%s_SQ_WAVES+SQ_BUSY_CYCLES+SQ_INSTS_VALU+SQ_ACTIVE_INST_VALU.log" % prefix)
    bl = [l for l in open(log).read().splitlines() if l.startswith("{")]
    jb = json.loads(bl[-1])
    r = jb["roofline"]
    pk = r["per_kernel"]["k_optimize<1> (host-visible rounds)"]
    share = (pk["algorithmic_bytes_per_launch"] * pk["launches"]) / (r["algorithmic_bytes_per_launch"] * r["launches"])
    passes_bulk_per_step = r["n_pass"] * share / (jb["steps"] * len(jb.get("repeats", [1])))
    a2, _ = (acc, regs) if prefix == "pmc" else collect(prefix)
    valu = sum(a2[k]["SQ_INSTS_VALU"][0] for k in a2 if family(k) == "k_optimize<1> (host-visible rounds)" and "SQ_INSTS_VALU" in a2[k]) / steps_all
    return valu, passes_bulk_per_step, valu / (passes_bulk_per_step / 64.0)
traffic["valu_wave_insts_per_wave_pass_by_plan"] = {}
for prefix, plan in (("pmc", PLAN["pmc"]), ("pmcd", PLAN["pmcd"])):
    try:
        valu, passes, per = valu_per_wave_pass(prefix, STEPS[prefix])
        traffic["valu_wave_insts_per_wave_pass_by_plan"][plan] = per
        lines += ["", "Bulk kernels (k_optimize<Lay<1,.>> + k_optimize_spec), %s: %.3e VALU wave-instructions per step, %.3e counted passes per step -> %.0f VALU wave-instructions per wavefront pass (%.1f per sample)"
                  % (plan, valu, passes, per, per / 25.0)]
    except Exception as e:
        lines += ["", "(no VALU-per-pass figure for %s: %r)" % (plan, e)]
# the same for the vector L1: TCP_TOTAL_CACHE_ACCESSES (one per lane of a gather) of the bulk kernels per wavefront pass --
# bench.py's `l1_gather` roof (the L1's rate on this access pattern: tools/ubench/gather_rate.hip) reads them here
def l1_per_wave_pass(prefix, steps_all):
    log = os.path.join(src, "%s_TCP_TOTAL_CACHE_ACCESSES_sum+TCP_TCC_READ_REQ_sum+TCP_PENDING_STALL_CYCLES_sum.log" % prefix)
    bl = [l for l in open(log).read().splitlines() if l.startswith("{")]
    jb = json.loads(bl[-1])
    r = jb["roofline"]
    pk = r["per_kernel"]["k_optimize<1> (host-visible rounds)"]
    share = (pk["algorithmic_bytes_per_launch"] * pk["launches"]) / (r["algorithmic_bytes_per_launch"] * r["launches"])
    passes_bulk_per_step = r["n_pass"] * share / (jb["steps"] * len(jb.get("repeats", [1])))
    a2, _ = (acc, regs) if prefix == "pmc" else collect(prefix)
    l1 = sum(a2[k]["TCP_TOTAL_CACHE_ACCESSES_sum"][0] for k in a2 if family(k) == "k_optimize<1> (host-visible rounds)" and "TCP_TOTAL_CACHE_ACCESSES_sum" in a2[k]) / steps_all
    return l1, passes_bulk_per_step, l1 / (passes_bulk_per_step / 64.0)
traffic["l1_accesses_per_wave_pass_by_plan"] = {}
for prefix, plan in (("pmc", PLAN["pmc"]), ("pmcd", PLAN["pmcd"])):
    try:
        l1, passes, per = l1_per_wave_pass(prefix, STEPS[prefix])
        traffic["l1_accesses_per_wave_pass_by_plan"][plan] = per
        lines += ["", "Bulk kernels, %s: %.3e L1 accesses per step -> %.0f per wavefront pass (%.1f per sample: a 64-lane gather is 64)" % (plan, l1, per, per / 25.0)]
    except Exception as e:
        lines += ["", "(no L1-accesses-per-pass figure for %s: %r)" % (plan, e)]
if PLAN["pmc"] in traffic["valu_wave_insts_per_wave_pass_by_plan"]:
    traffic["valu_wave_insts_per_wave_pass"] = traffic["valu_wave_insts_per_wave_pass_by_plan"][PLAN["pmc"]]
if traffic["plans"]:
    json.dump(traffic, open(os.path.join(dst, "%s_traffic.json" % tag), "w"), indent=1)
for n in ("1thread", "default", "driver", "1thread_5steps"):
    f = os.path.join(src, "bench_%s.json" % n)
    if os.path.exists(f) and os.path.getsize(f):
        j = json.loads(open(f).read().strip().splitlines()[-1])
        lines += ["", "bench (%s): %.1f %s, %.2f ms/step, roofline %s" % (n, j["value"], j["unit"], j["ms_per_step"], json.dumps(j["roofline"]))]
open(os.path.join(dst, "%s_pmc.md" % tag), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
