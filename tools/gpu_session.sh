#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): small host-visible rounds as ONE launch of the general kernel.
TAG=${1:-r3w}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
show() {
  python - $1 $2 <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); t = d['roofline']['per_kernel']['k_tail + k_front (tail rounds)']; b = d['roofline']['per_kernel']['k_optimize<1> (host-visible rounds)']
    print('%-14s' % sys.argv[2], round(d['value'], 1), 'maps/s', d['config']['host_threads_per_gpu'], 'thr', 'ms/step', round(d['ms_per_step'], 2), 'bulk ms', round(b['avg_launch_ms'] * b['launches'] / d['steps'], 2), b['launches'], 'k_tail', round(t['k_tail_ms'] / d['steps'], 2), t.get('k_tail_launches'), 'k_front', round(t['k_front_ms'] / d['steps'], 2))
except Exception as e:
    print(sys.argv[2], 'failed', e)
PY
}
one() { L=$1; shift; env "$@" timeout -s KILL 120 python bench.py --no-cpu-baseline --no-one-call --streams 1 --steps-per-call 1 --steps 10 --warmup 2 2>/dev/null > $OUT/b1_$L.json; show $OUT/b1_$L.json one_$L; }
drv() { L=$1; shift; env "$@" timeout -s KILL 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-one-call 2>/dev/null > $OUT/drv_$L.json; show $OUT/drv_$L.json drv_$L; }
one ol0
one ol50k MI_DMRECON_ONE_LAUNCH=50000
one ol100k MI_DMRECON_ONE_LAUNCH=100000
one ol200k MI_DMRECON_ONE_LAUNCH=200000
one ol1M MI_DMRECON_ONE_LAUNCH=100000000
drv ol0
drv ol100k MI_DMRECON_ONE_LAUNCH=100000
for N in 3; do for L in 0 100000 100000000; do
  MI_DMRECON_ONE_LAUNCH=$L timeout -s KILL 100 python tools/trace_c3.py C3 $N 2>&1 | grep -E "phase (seeds|phase)|wall" | sed 's/\[mi_dmrecon\] phase//' | tr '\n' ' ' | cut -c1-200 | sed "s/^/lone $N views, one launch below $L: /"; echo
done; done
