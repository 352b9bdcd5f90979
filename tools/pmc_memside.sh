#!/bin/bash
# GPU box: memory-side counters of the optimise kernels (small counter groups, one per run, kernel-trace only;
# every run under its own timeout: an over-subscribed group makes rocprofv3 abort and then hang in finalisation)
export TMPDIR=/tmp
OUT=gpurun_out/pmc_mem; mkdir -p $OUT; : > $OUT/summary.txt
BP="python bench.py --steps 1 --warmup 1 --streams 1 --steps-per-call 1 --no-cpu-baseline"
i=0
for C in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
         "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
         "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES" \
         "TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -s KILL 90 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pm$i -o b -- $BP > $OUT/run$i.log 2>&1
  python - /tmp/pm$i/b_counter_collection.csv >> $OUT/summary.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
try:
    for r in csv.DictReader(open(sys.argv[1])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
except Exception as e:
    print("ERR", e)
for k in ("k_optimize<1>", "k_tail"):
    for c, (v, n) in acc[k].items():
        print("%-16s %-40s per-launch %.5g  (n=%d)" % (k, c, v / max(n, 1), n))
PY
done
cat $OUT/summary.txt
