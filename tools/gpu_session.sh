#!/bin/bash
# Runs ON THE GPU BOX (through gpurun).  Merge policy sweep on the default bench plan.
TAG=${1:-s38}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
run() { N=$1; shift
  env "$@" timeout -s KILL 300 python bench.py --no-cpu-baseline $EXTRA > $OUT/$N.json 2> $OUT/$N.err
  python - $OUT/$N.json $N <<'PY'
import sys, json
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d['value'], 1), 'maps/s;', d['config']['library_batches'], 'batches of', d['config']['views_per_library_batch'], 'views; bulk frac', round(d['roofline']['per_kernel']['k_optimize<1> (host-visible rounds)']['frac'], 4))
PY
}
EXTRA="" run base_1
EXTRA="" run run3 MI_DMRECON_MERGE_RUNNING=3
EXTRA="" run win2000 MI_DMRECON_MERGE_WINDOW_US=2000
EXTRA="" run run1_win0 MI_DMRECON_MERGE_RUNNING=1 MI_DMRECON_MERGE_WINDOW_US=0
EXTRA="--streams 12" run t12
EXTRA="--streams 3" run t3
EXTRA="" run base_2
EXTRA="" run nomerge MI_DMRECON_MERGE_CALLS=0
