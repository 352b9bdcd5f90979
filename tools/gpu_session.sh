#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): whole GPU suite after the team front / hand-over rule / shim changes, the drop-in
# app's timing, the bench plans.
TAG=${1:-r3p}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -s KILL 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
MI_DMRECON_TRACE=1 timeout -s KILL 300 python tools/app_c3_timing.py 2>&1 | cut -c1-200 | tail -60 > $OUT/app_trace.txt
tail -45 $OUT/app_trace.txt
show() {
  python - $1 $2 <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); t = d['roofline']['per_kernel']['k_tail + k_front (tail rounds)']; b = d['roofline']['per_kernel']['k_optimize<1> (host-visible rounds)']
    print('%-14s' % sys.argv[2], round(d['value'], 1), 'maps/s', d['config']['host_threads_per_gpu'], 'thr', 'ms/step', round(d['ms_per_step'], 2), 'bulk ms', round(b['avg_launch_ms'] * b['launches'] / d['steps'], 2), 'k_tail', round(t['k_tail_ms'] / d['steps'], 2), t.get('k_tail_launches'), 'k_front', round(t['k_front_ms'] / d['steps'], 2), 'rounds', t.get('k_front_rounds_slowest_view'), 'att', t.get('k_front_attempts'), 'one_call', (d.get('one_call') or {}).get('value'))
except Exception as e:
    print(sys.argv[2], 'failed', e)
PY
}
drv() { L=$1; shift; env "$@" timeout -s KILL 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null > $OUT/drv_$L.json; show $OUT/drv_$L.json drv_$L; }
dfl() { L=$1; shift; env "$@" timeout -s KILL 300 python bench.py --no-cpu-baseline --no-one-call 2>/dev/null > $OUT/dfl_$L.json; show $OUT/dfl_$L.json dfl_$L; }
sml() { L=$1; shift; env "$@" timeout -s KILL 200 python bench.py --no-cpu-baseline --no-one-call --streams 6 --steps-per-call 1 --steps 60 --warmup 3 2>/dev/null > $OUT/sml_$L.json; show $OUT/sml_$L.json sml_$L; }
one() { L=$1; shift; env "$@" timeout -s KILL 120 python bench.py --no-cpu-baseline --no-one-call --streams 1 --steps-per-call 1 --steps 10 --warmup 2 2>/dev/null > $OUT/b1_$L.json; show $OUT/b1_$L.json one_$L; }
drv dflt_1
drv dflt_2
dfl dflt
sml dflt
sml all MI_DMRECON_FRONT=1000000
one dflt
timeout -s KILL 200 python bench.py --no-cpu-baseline --no-one-call --streams 1 --steps-per-call 5 --steps 20 --warmup 5 2>/dev/null > $OUT/b1_100.json; show $OUT/b1_100.json one_thread_100views
