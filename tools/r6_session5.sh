#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): round 6, fifth session -- the first follow-up list of a large round in list order (A/B), the
# view-selection tables per part of a bundle (the distinct-scenes variant), the GPU suite, the driver's line.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=gpurun_out/r6e
mkdir -p $O
NOX="--no-cpu-baseline --no-one-call --distinct-scenes 0 --no-seed-variant"
timeout -s KILL 600 python -m pytest tests/test_gpu_parity.py -x -q -k "batch or speculative or sparse or seed or deterministic or front_kernel_equals" > $O/pytest_quick.log 2>&1; tail -3 $O/pytest_quick.log
for V in "MI_DMRECON_FOLLOW_ORDERED=1" "MI_DMRECON_FOLLOW_ORDERED=0" "MI_DMRECON_FOLLOW_ORDERED=1" "MI_DMRECON_FOLLOW_ORDERED=0"; do
  T=$(echo $V | tr ' =' '__')_$RANDOM
  env $V MI_BENCH_REGION_LOG=1 timeout -s KILL 200 python bench.py --steps 20 --warmup 5 --repeats 3 $NOX > $O/bench_$T.json 2> $O/bench_$T.err
  echo "$V: $(python -c "import json,sys; d=json.loads(open('$O/bench_$T.json').read().strip().splitlines()[-1]); print(round(d['value'],1), [round(x) for x in d['repeats']], 'clock', d['roofline']['shader_clock_mhz_measured'])")"
  grep region $O/bench_$T.err | tail -1
done
cd /tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --output-format csv -d $R/$O/kt_ord -o bench -- python $R/bench.py --steps 20 --warmup 5 --repeats 2 $NOX > $R/$O/kt_ord.log 2>&1
cd $R
for K in kt_ord; do
  F=$(find $O/$K -name "*kernel_trace.csv" | head -1)
  [ -n "$F" ] && python tools/trace_regions.py $F 20 > $O/$K.json 2> $O/$K.err
  [ -n "$F" ] && gzip -c $F > $O/$K.trace.csv.gz && rm -f $F
  python -c "import json; d=json.load(open('$O/$K.json')); print('$K', json.dumps(d['all_regions']['families_ms_per_step'])[:700])"
done
( timeout -s KILL 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "rc $?" >> $O/pytest_gpu.log ); tail -5 $O/pytest_gpu.log
MI_DMRECON_TRACE= MI_BENCH_REGION_LOG=1 timeout -s KILL 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
grep "region" $O/bench_driver.err | tail -4
python - <<PY
import json
d = json.loads(open("$O/bench_driver.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value", round(d["value"], 1), [round(x) for x in d["repeats"]], "frac", round(r["frac"], 3), "bulk", round(r["bulk_kernel_frac"], 3), "clock", r["shader_clock_mhz_measured"])
print("one_call", d["one_call"]["ms_per_call"], d["one_call"]["ms_front_kernel"], d["one_call"]["ms_bulk_kernel"])
print("distinct", d["config"]["distinct_scenes_variant"]["value"], d["config"]["distinct_scenes_variant"]["repeats"], "seedvar", d["config"]["all_seeds_propagate_variant"]["value"])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["kind"], d.get("parity", {}).get("within_bounds"), d.get("parity", {}).get("min_fill_iou"))
PY
du -sh $O
