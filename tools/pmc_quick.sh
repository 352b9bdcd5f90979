#!/bin/bash
# GPU box: one small PMC group + kernel stats for the bulk kernel (each run under its own timeout)
export TMPDIR=/tmp
BP="python bench.py --steps 1 --warmup 1 --streams 1 --steps-per-call 1 --no-cpu-baseline"
timeout -s KILL 90 rocprofv3 --kernel-trace --pmc ${PMC:-TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum} --output-format csv -d /tmp/pq -o b -- $BP > /tmp/pq.log 2>&1
python - /tmp/pq/b_counter_collection.csv <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k in ("k_optimize<1>", "k_tail"):
    for c, (v, n) in acc[k].items():
        print("%-16s %-40s total %.5g per-launch %.5g  (n=%d)" % (k, c, v, v / max(n, 1), n))
PY
timeout -s KILL 90 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps -o b -- python bench.py --steps 6 --warmup 1 --streams 1 --steps-per-call 1 --no-cpu-baseline > /tmp/ps.log 2>&1
cut -c1-110 /tmp/ps/b_kernel_stats.csv | head -4
