#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): SQ counters of the optimise kernels at a lone 20-view call (FRONT on and off).
TAG=${1:-r3d}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BP="python bench.py --steps 2 --warmup 1 --streams 1 --steps-per-call 1 --no-cpu-baseline --no-one-call"
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD" "SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC"; do
  D=$OUT/pmc_$(echo $C | tr ' ' '+')
  MI_DMRECON_FRONT=1000000 timeout -s KILL 120 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D -o bench -- $BP > $D.log 2>&1
done
python - $OUT <<'PY'
import csv, sys, collections, glob, os
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(sys.argv[1], "pmc_*", "bench_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(acc):
    if not (k.startswith("mi_fw5::k_optimize") or k.startswith("mi_fw5::k_tail") or k.startswith("mi_fw5::k_front") or k.startswith("k_")):
        pass
    print(k)
    for c, (v, n) in sorted(acc[k].items()):
        print("    %-28s total %.6g  per launch %.6g (n=%d)" % (c, v, v / max(n, 1), n))
PY
find $OUT -name "*_kernel_trace.csv" -delete
