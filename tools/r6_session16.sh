#!/bin/bash
# Runs ON THE GPU BOX: round 6 -- the round driver's thresholds again under the column-pair elements (no rebuild: environment
# switches): hand-over size, single-attempt follow-up launches, speculative-round threshold; driver's plan and lone calls.
export TMPDIR=/tmp
O=gpurun_out/r6q
mkdir -p $O
NOX="--no-cpu-baseline --no-one-call --distinct-scenes 0 --no-seed-variant"
run() {   # tag, env...
  T=$1; shift
  env "$@" MI_BENCH_REGION_LOG=1 timeout -s KILL 200 python bench.py --steps 20 --warmup 5 --repeats 3 $NOX > $O/bench_$T.json 2> $O/bench_$T.err
  V=$(python -c "import json,sys; d=json.loads(open('$O/bench_$T.json').read().strip().splitlines()[-1]); print(round(d['value'],1), [round(x) for x in d['repeats']])")
  env "$@" timeout -s KILL 200 python bench.py --streams 1 --steps-per-call 1 --steps 20 --warmup 3 --repeats 3 $NOX > $O/lone_$T.json 2> $O/lone_$T.err
  L=$(python -c "import json,sys; d=json.loads(open('$O/lone_$T.json').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2))")
  echo "$T: driver $V | lone $L ms | $(grep region $O/bench_$T.err | tail -1 | sed 's/.*kernels/kernels/')"
}
run base A=1
run ho160 MI_DMRECON_VIEW_HANDOVER=160
run ho640 MI_DMRECON_VIEW_HANDOVER=640
run sf2 MI_DMRECON_SINGLE_FOLLOW=2
run sf3 MI_DMRECON_SINGLE_FOLLOW=3
run ff1 MI_DMRECON_FAST_FOLLOW=1
run spec200 MI_DMRECON_SPEC_ROUNDS=200000
run spec800 MI_DMRECON_SPEC_ROUNDS=800000
run one50 MI_DMRECON_ONE_LAUNCH=50000
run one200 MI_DMRECON_ONE_LAUNCH=200000
run base2 A=1
