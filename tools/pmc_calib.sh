#!/bin/bash
# GPU box: FETCH_SIZE / WRITE_SIZE of kernels with known byte counts (tools/ubench/pmc_calib.hip), one counter per
# run (kernel-trace only).  Output: gpurun_out/pmc_calib/summary.txt -> profiles/r2_pmc_calibration.md
export TMPDIR=/tmp
OUT=gpurun_out/pmc_calib; mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/ubench/pmc_calib.hip -o /tmp/pmc_calib || exit 1
/tmp/pmc_calib > $OUT/known.txt; cat $OUT/known.txt
for C in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  D=/tmp/pc_$(echo $C | tr ' ' '+')
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D -o c -- /tmp/pmc_calib > $OUT/run_$(echo $C | tr ' ' '+').log 2>&1
  python - $D/c_counter_collection.csv <<'PY' | tee -a $OUT/summary.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
try:
    for r in csv.DictReader(open(sys.argv[1])):
        k = r["Kernel_Name"].split("(")[0]
        a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
except Exception as e:
    print("ERR", e)
for k in sorted(acc):
    for c, (v, n) in acc[k].items():
        print("%-14s %-24s per-launch %.6g  (n=%d)" % (k, c, v / max(n, 1), n))
PY
done
