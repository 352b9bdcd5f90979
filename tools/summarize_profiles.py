"""Turns gpurun_out/<tag>/ (written by tools/collect_profiles.sh on the GPU box) into the committed summaries
under profiles/: kernel stats CSVs (rocprofv3 --kernel-trace --stats), bench JSON lines and a PMC table."""
import csv
import collections
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
src = os.path.join("gpurun_out", tag)
dst = "profiles"
for n in ("1thread", "default"):
    f = os.path.join(src, "bench_%s.json" % n)
    if os.path.exists(f) and os.path.getsize(f):
        shutil.copy(f, os.path.join(dst, "%s_bench_%s.json" % (tag, n)))
for s in ("s1", "s3"):
    f = os.path.join(src, s, "bench_kernel_stats.csv")
    if os.path.exists(f):
        shutil.copy(f, os.path.join(dst, "%s_kernel_stats_%s.csv" % (tag, "1thread" if s == "s1" else "6threads")))


def short(name):
    return name.split("(")[0].replace("void ", "")


acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
regs = {}
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    f = os.path.join(d, "bench_counter_collection.csv")
    if not os.path.exists(f):
        continue
    for row in csv.DictReader(open(f)):
        k = short(row["Kernel_Name"])
        a = acc[k][row["Counter_Name"]]
        a[0] += float(row["Counter_Value"]); a[1] += 1
        regs[k] = (row["VGPR_Count"], row["Accum_VGPR_Count"], row["SGPR_Count"], row["LDS_Block_Size"], row["Scratch_Size"])
lines = ["# %s — PMC counters of the C3 bench (rocprofv3 --pmc, one counter group per run; see tools/collect_profiles.sh)" % tag, "",
         "Per launch averages over `python bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline` (3 reconstructions",
         "of the 20-view scene).  FETCH_SIZE / WRITE_SIZE are reported in KB and shown here in MB; no x2 correction is",
         "applied (the guide's doubling concerns 16 B/lane streams, these kernels gather 4-8 B per lane).", "",
         "| kernel | launches | FETCH MB | WRITE MB | L2 hit % | VALU-active % of wave cycles | wave-cycles/VALU inst | wait-any % | L1 accesses | L1->L2 read req | L1 pending-stall % of L1 busy | VGPR/AGPR/SGPR | LDS B | scratch B |",
         "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
for k in sorted(acc, key=lambda k: -acc[k].get("FETCH_SIZE", [0, 0])[0]):
    c = acc[k]
    if k.startswith("__amd"):
        continue
    def avg(n):
        return c[n][0] / c[n][1] if n in c and c[n][1] else None
    n = max(v[1] for v in c.values())
    hit, miss = avg("TCC_HIT_sum"), avg("TCC_MISS_sum")
    busy, act = avg("SQ_BUSY_CYCLES"), avg("SQ_ACTIVE_INST_VALU")
    wc, valu, wany = avg("SQ_WAVE_CYCLES"), avg("SQ_INSTS_VALU"), avg("SQ_WAIT_INST_ANY")
    f = lambda v, s=1.0, fmt="%.3f": "-" if v is None else fmt % (v * s)
    l1a, l1r, l1p, l1g = avg("TCP_TOTAL_CACHE_ACCESSES_sum"), avg("TCP_TCC_READ_REQ_sum"), avg("TCP_PENDING_STALL_CYCLES_sum"), avg("TCP_GATE_EN1_sum")
    lines.append("| `%s` | %d | %s | %s | %s | %s | %s | %s | %s | %s | %s | %s | %s | %s |" % (
        k, n, f(avg("FETCH_SIZE"), 1 / 1024), f(avg("WRITE_SIZE"), 1 / 1024),
        "-" if hit is None or hit + miss == 0 else "%.1f" % (100 * hit / (hit + miss)),
        "-" if not wc or act is None else "%.1f" % (100 * act / wc),
        "-" if not valu or wc is None else "%.2f" % (wc / valu),
        "-" if not wc or wany is None else "%.1f" % (100 * wany / wc),
        "-" if l1a is None else "%.3g" % l1a, "-" if l1r is None else "%.3g" % l1r,
        "-" if not l1g or l1p is None else "%.1f" % (100 * l1p / l1g),
        "/".join(regs[k][:3]), regs[k][3], regs[k][4]))
# HBM-side traffic of the optimise kernels per launch (all layouts pooled, like bench.py's `achieved`)
tb, tl = 0.0, 0
for k in acc:
    if (k.startswith("k_optimize") or k == "k_tail") and "FETCH_SIZE" in acc[k] and "WRITE_SIZE" in acc[k]:
        tb += (acc[k]["FETCH_SIZE"][0] + acc[k]["WRITE_SIZE"][0]) * 1024.0
        tl += acc[k]["FETCH_SIZE"][1]
if tl:
    json.dump({"kernel": "k_optimize", "bytes_per_launch": tb / tl, "launches": tl,
               "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), profiles/%s_pmc.md" % tag},
              open(os.path.join(dst, "%s_traffic.json" % tag), "w"))
    lines += ["", "optimise kernels (k_optimize<1>, k_optimize<16>, k_tail pooled, as bench.py counts launches): %.2f MB of HBM-side traffic per launch over %d launches" % (tb / tl / 1e6, tl)]
for n in ("1thread", "default"):
    f = os.path.join(src, "bench_%s.json" % n)
    if os.path.exists(f) and os.path.getsize(f):
        j = json.loads(open(f).read().strip().splitlines()[-1])
        lines += ["", "bench (%s): %.1f %s, %.2f ms/step, roofline %s" % (n, j["value"], j["unit"], j["ms_per_step"], json.dumps(j["roofline"]))]
open(os.path.join(dst, "%s_pmc.md" % tag), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
