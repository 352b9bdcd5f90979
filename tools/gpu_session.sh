#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the bulk turn (one batch at a time in the host-visible rounds) against the
# plans of the bench: the driver's, the default, many small calls, the lone call.
TAG=${1:-r3m}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
show() {
  python - $1 $2 <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); t = d['roofline']['per_kernel']['k_tail + k_front (tail rounds)']; b = d['roofline']['per_kernel']['k_optimize<1> (host-visible rounds)']
    print('%-14s' % sys.argv[2], round(d['value'], 1), 'maps/s', d['config']['host_threads_per_gpu'], 'thr', d['config']['library_batches'], 'batches', d['config'].get('library_batch_log')[:8], 'bulk frac', round(d['roofline']['bulk_kernel_frac'], 4),
          'bulk ms', round(b['avg_launch_ms'] * b['launches'] / d['steps'], 2), 'k_tail', round(t['k_tail_ms'] / d['steps'], 2), 'k_front', round(t['k_front_ms'] / d['steps'], 2))
except Exception as e:
    print(sys.argv[2], 'failed', e)
PY
}
drv() { L=$1; shift; env "$@" timeout -s KILL 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-one-call 2>/dev/null > $OUT/drv_$L.json; show $OUT/drv_$L.json drv_$L; }
dfl() { L=$1; shift; env "$@" timeout -s KILL 300 python bench.py --no-cpu-baseline --no-one-call 2>/dev/null > $OUT/dfl_$L.json; show $OUT/dfl_$L.json dfl_$L; }
sml() { L=$1; shift; env "$@" timeout -s KILL 200 python bench.py --no-cpu-baseline --no-one-call --streams 6 --steps-per-call 1 --steps 60 --warmup 3 2>/dev/null > $OUT/sml_$L.json; show $OUT/sml_$L.json sml_$L; }
one() { L=$1; shift; env "$@" timeout -s KILL 120 python bench.py --no-cpu-baseline --no-one-call --streams 1 --steps-per-call 1 --steps 10 --warmup 2 2>/dev/null > $OUT/b1_$L.json; show $OUT/b1_$L.json one_$L; }
for R in 1 2; do
drv noturn_$R MI_DMRECON_BULK_TURN=0
drv turn_$R
drv turn_nomerge_$R MI_DMRECON_MERGE_CALLS=0
drv turn_run4_v128_$R MI_DMRECON_MAX_RUNNING=4 MI_DMRECON_MERGE_VIEWS=128
drv turn_run3_v200_$R MI_DMRECON_MAX_RUNNING=3 MI_DMRECON_MERGE_VIEWS=200
drv turn_run4_$R MI_DMRECON_MAX_RUNNING=4
done
dfl noturn MI_DMRECON_BULK_TURN=0
dfl turn
dfl turn_run4_v128 MI_DMRECON_MAX_RUNNING=4 MI_DMRECON_MERGE_VIEWS=128
dfl turn_nomerge MI_DMRECON_MERGE_CALLS=0
sml noturn MI_DMRECON_BULK_TURN=0
sml turn
sml turn_run4_v128 MI_DMRECON_MAX_RUNNING=4 MI_DMRECON_MERGE_VIEWS=128
sml turn_run4_v64 MI_DMRECON_MAX_RUNNING=4 MI_DMRECON_MERGE_VIEWS=64
sml turn_nomerge MI_DMRECON_MERGE_CALLS=0
one turn
