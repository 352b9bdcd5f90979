#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): (1) the driver's exact command as the FIRST GPU process of the box, three times --
# what the batches looked like; (2) merge policies; (3) follow-up launches in the latency layout; (4) lone calls of 3..20 views.
TAG=${1:-r3l}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
show() {
  python - $1 $2 <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); t = d['roofline']['per_kernel']['k_tail + k_front (tail rounds)']; b = d['roofline']['per_kernel']['k_optimize<1> (host-visible rounds)']
    print('%-14s' % sys.argv[2], round(d['value'], 1), 'maps/s', d['config']['host_threads_per_gpu'], 'thr', d['config']['library_batches'], 'batches', d['config'].get('library_batch_log'), 'bulk frac', round(d['roofline']['bulk_kernel_frac'], 4),
          'bulk ms', round(b['avg_launch_ms'] * b['launches'] / d['steps'], 2), 'k_tail', round(t['k_tail_ms'] / d['steps'], 2), 'k_front', round(t['k_front_ms'] / d['steps'], 2))
except Exception as e:
    print(sys.argv[2], 'failed', e)
PY
}
for R in 1 2 3; do
  timeout -s KILL 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/official_$R.json 2>/dev/null; show $OUT/official_$R.json official_$R
done
drv() { L=$1; shift; env "$@" timeout -s KILL 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-one-call 2>/dev/null > $OUT/drv_$L.json; show $OUT/drv_$L.json $L; }
one() { L=$1; shift; env "$@" timeout -s KILL 120 python bench.py --no-cpu-baseline --no-one-call --streams 1 --steps-per-call 1 --steps 10 --warmup 2 2>/dev/null > $OUT/b1_$L.json; show $OUT/b1_$L.json $L; }
for R in 1 2; do
drv nomerge_$R MI_DMRECON_MERGE_CALLS=0
drv win20ms_$R MI_DMRECON_MERGE_WINDOW_US=20000
drv dflt_$R
done
one fl0 MI_DMRECON_FOLLOW_LAT=0
one fl40k MI_DMRECON_FOLLOW_LAT=40000
one fl100k MI_DMRECON_FOLLOW_LAT=100000
one fl200k MI_DMRECON_FOLLOW_LAT=200000
drv fl100k MI_DMRECON_FOLLOW_LAT=100000
for N in 3 5 10 20; do timeout -s KILL 100 python tools/trace_c3.py C3 $N 2>&1 | grep -E "phase|wall" | tr '\n' ' ' | sed "s/^/lone call of $N views: /"; echo; done
