#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): round 5, session C -- XCD-aware work-list ranges (every XCD an eighth of a round's list)
# against block order (-DMI_XCDS=1) and the transposed lane layout, on ONE scene and on 20 DISTINCT scenes; the parity tests on
# the product build; instruction-cache counters of the bulk kernels.
export TMPDIR=/tmp
O=gpurun_out/r5c
mkdir -p $O
line() { python - "$1" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
oc = j.get("one_call") or {}
d = j["config"].get("distinct_scenes_variant") or {}
print("%s: value %.1f %s | bulk frac %.3f | one_call %.2f ms (bulk %.2f front %.2f) | distinct %.1f (bulk %.2f ms/step, front %.2f)" % (
      sys.argv[1], j["value"], [round(v) for v in j["repeats"]], j["roofline"]["bulk_kernel_frac"], oc.get("ms_per_call", 0),
      oc.get("ms_bulk_kernel", 0), oc.get("ms_front_kernel", 0), d.get("value", 0), d.get("ms_bulk_kernel_per_step", 0), d.get("ms_front_kernel_per_step", 0)))
PY
}
timeout -s KILL 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
AB="--steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --distinct-scenes 20 --one-call-n 20"
MI_BENCH_REGION_LOG=1 timeout -s KILL 300 python bench.py $AB > $O/bench_base.json 2> $O/bench_base.err
line $O/bench_base.json; grep "^region" $O/bench_base.err | sed -n '3p;6p'
for V in nox tr; do
  L=build/libmi_dmrecon_$V.so
  [ -f $L ] || continue
  MI_BENCH_REGION_LOG=1 MI_DMRECON_LIB=$PWD/$L timeout -s KILL 300 python bench.py $AB > $O/bench_$V.json 2> $O/bench_$V.err
  line $O/bench_$V.json; grep "^region" $O/bench_$V.err | sed -n '3p;6p'
done
MI_BENCH_REGION_LOG=1 timeout -s KILL 300 python bench.py $AB > $O/bench_base2.json 2> $O/bench_base2.err
line $O/bench_base2.json
# instruction cache and L2 of the bulk kernels at the driver's plan (one counter group per run)
rocprofv3 -L 2>/dev/null | grep -i -E "ICACHE|IFETCH|INST_CACHE" | head -20 > $O/counters_icache.txt; head -12 $O/counters_icache.txt
R=$PWD; cd /tmp
BQ="python $R/bench.py --steps 20 --warmup 1 --repeats 1 --no-cpu-baseline --no-one-call --distinct-scenes 0"
for C in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_IFETCH"; do
  D=$R/$O/pmcd_$(echo $C | tr ' ' '+')
  timeout -s KILL 150 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D -o bench -- $BQ > $D.log 2>&1
  python - "$D" <<'PY'
import csv, collections, glob, os, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        if "k_optimize" in k or "k_front" in k:
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
for k, c in acc.items():
    print(k[:70], {n: "%.4g" % v for n, v in c.items()})
PY
done
cd $R; find $O -name "*_kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +2M -delete; du -sh $O
