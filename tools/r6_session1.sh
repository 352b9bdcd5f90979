#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): round 6, first measurement session -- the baseline of the box, the per-launch trace of a
# 400-view batch, a kernel trace cut to the timed regions, and the merge-policy experiments (one batch / two / four side by side).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=gpurun_out/r6a
mkdir -p $O
NOX="--no-cpu-baseline --no-one-call --distinct-scenes 0 --no-seed-variant"
MI_BENCH_REGION_LOG=1 timeout -s KILL 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
cut -c1-400 $O/bench_driver.json; grep region $O/bench_driver.err | tail -5
# per-launch trace of one 400-view batch
MI_DMRECON_TRACE=1 timeout -s KILL 200 python bench.py --steps 20 --warmup 1 --repeats 1 $NOX > $O/trace400.json 2> $O/trace400.err
# merge-policy experiments
for V in "MI_DMRECON_MERGE_SPLIT=2" "MI_DMRECON_MERGE_CALLS=0" "MI_DMRECON_MERGE_SPLIT=2 GPU_MAX_HW_QUEUES=16" "MI_DMRECON_MERGE_SPLIT=4 MI_DMRECON_MERGE_RUNNING=4"; do
  T=$(echo $V | tr ' =' '__')
  env $V MI_BENCH_REGION_LOG=1 timeout -s KILL 200 python bench.py --steps 20 --warmup 5 --repeats 3 $NOX > $O/bench_$T.json 2> $O/bench_$T.err
  echo "$V: $(python -c "import json,sys; d=json.loads(open('$O/bench_$T.json').read().strip().splitlines()[-1]); print(round(d['value'],1), [round(x) for x in d['repeats']], d['config']['calls_per_library_batch_by_region'])")"
  grep region $O/bench_$T.err | tail -4
done
# kernel traces cut to the timed regions: the default plan and the split plan
cd /tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --output-format csv -d $R/$O/kt_default -o bench -- python $R/bench.py --steps 20 --warmup 5 --repeats 2 $NOX > $R/$O/kt_default.log 2>&1
MI_DMRECON_MERGE_SPLIT=2 timeout -s KILL 240 rocprofv3 --kernel-trace --output-format csv -d $R/$O/kt_split2 -o bench -- python $R/bench.py --steps 20 --warmup 5 --repeats 2 $NOX > $R/$O/kt_split2.log 2>&1
cd $R
for K in kt_default kt_split2; do
  F=$(find $O/$K -name "*kernel_trace.csv" | head -1)
  [ -n "$F" ] && python tools/trace_regions.py $F 20 > $O/$K.json 2> $O/$K.err
  # keep one region's worth of the raw trace for the timeline (the second timed region), compressed
  [ -n "$F" ] && gzip -c $F > $O/$K.trace.csv.gz && rm -f $F
  python -c "import json; d=json.load(open('$O/$K.json')); print('$K', json.dumps(d['all_regions'])[:1500])"
done
du -sh $O
