#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): round 6, second session -- the GPU suite on the tree with the per-template counters, the
# driver's line with the shader clock, and the follow-up launches in the FAST kernel (MI_DMRECON_FAST_FOLLOW) A/B.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=gpurun_out/r6b
mkdir -p $O
NOX="--no-cpu-baseline --no-one-call --distinct-scenes 0 --no-seed-variant"
( timeout -s KILL 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "rc $?" >> $O/pytest_gpu.log ); tail -5 $O/pytest_gpu.log
MI_BENCH_REGION_LOG=1 timeout -s KILL 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
tail -c 300 $O/bench_driver.err
python - <<PY
import json
d = json.loads(open("$O/bench_driver.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value", round(d["value"], 1), [round(x) for x in d["repeats"]], "frac", round(r["frac"], 3), "bulk", round(r["bulk_kernel_frac"], 3))
print("valu", r["secondary_roofs"]["valu_issue"])
print("templates", json.dumps(r["per_kernel_template"]))
print("latency rounds", r["latency_layout_rounds"])
print("one_call", d["one_call"]["ms_per_call"], d["one_call"]["ms_front_kernel"], d["one_call"]["ms_bulk_kernel"])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["kind"], d.get("parity", {}).get("within_bounds"), d.get("parity", {}).get("min_fill_iou"))
PY
for V in "MI_DMRECON_FAST_FOLLOW=0" "MI_DMRECON_FAST_FOLLOW=1" "MI_DMRECON_FAST_FOLLOW=2" "MI_DMRECON_FAST_FOLLOW=0" "MI_DMRECON_FAST_FOLLOW=1"; do
  T=$(echo $V | tr ' =' '__')_$RANDOM
  env $V MI_BENCH_REGION_LOG=1 timeout -s KILL 200 python bench.py --steps 20 --warmup 5 --repeats 3 $NOX > $O/bench_$T.json 2> $O/bench_$T.err
  echo "$V: $(python -c "import json,sys; d=json.loads(open('$O/bench_$T.json').read().strip().splitlines()[-1]); print(round(d['value'],1), [round(x) for x in d['repeats']], 'clock', d['roofline']['secondary_roofs']['valu_issue'])")"
  grep region $O/bench_$T.err | tail -2
done
cd /tmp
MI_DMRECON_FAST_FOLLOW=1 timeout -s KILL 240 rocprofv3 --kernel-trace --output-format csv -d $R/$O/kt_ff1 -o bench -- python $R/bench.py --steps 20 --warmup 5 --repeats 2 $NOX > $R/$O/kt_ff1.log 2>&1
cd $R
for K in kt_ff1; do
  F=$(find $O/$K -name "*kernel_trace.csv" | head -1)
  [ -n "$F" ] && python tools/trace_regions.py $F 20 > $O/$K.json 2> $O/$K.err
  [ -n "$F" ] && gzip -c $F > $O/$K.trace.csv.gz && rm -f $F
  python -c "import json; d=json.load(open('$O/$K.json')); print('$K', json.dumps(d['all_regions'])[:1200])"
done
du -sh $O
