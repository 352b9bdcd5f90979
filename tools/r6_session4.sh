#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): round 6, fourth session -- what neighbouring patches in a wavefront are worth: the first-attempt
# launches with their entries / their wavefront units in a scrambled order (MI_DMRECON_DEBUG_SCRAMBLE).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=gpurun_out/r6d
mkdir -p $O
NOX="--no-cpu-baseline --no-one-call --distinct-scenes 0 --no-seed-variant"
for V in "MI_DMRECON_DEBUG_SCRAMBLE=0" "MI_DMRECON_DEBUG_SCRAMBLE=1" "MI_DMRECON_DEBUG_SCRAMBLE=2" "MI_DMRECON_DEBUG_SCRAMBLE=0"; do
  T=$(echo $V | tr ' =' '__')_$RANDOM
  env $V MI_BENCH_REGION_LOG=1 timeout -s KILL 200 python bench.py --steps 20 --warmup 5 --repeats 3 $NOX > $O/bench_$T.json 2> $O/bench_$T.err
  echo "$V: $(python -c "import json,sys; d=json.loads(open('$O/bench_$T.json').read().strip().splitlines()[-1]); print(round(d['value'],1), [round(x) for x in d['repeats']], 'clock', d['roofline']['shader_clock_mhz_measured'])")"
  grep region $O/bench_$T.err | tail -1
done
cd /tmp
for S in 1 2; do
MI_DMRECON_DEBUG_SCRAMBLE=$S timeout -s KILL 240 rocprofv3 --kernel-trace --output-format csv -d $R/$O/kt_scr$S -o bench -- python $R/bench.py --steps 20 --warmup 5 --repeats 1 $NOX > $R/$O/kt_scr$S.log 2>&1
done
cd $R
for K in kt_scr1 kt_scr2; do
  F=$(find $O/$K -name "*kernel_trace.csv" | head -1)
  [ -n "$F" ] && python tools/trace_regions.py $F 20 > $O/$K.json 2> $O/$K.err
  [ -n "$F" ] && rm -f $F
  python -c "import json; d=json.load(open('$O/$K.json')); print('$K', json.dumps(d['all_regions']['families_ms_per_step'])[:600])"
done
