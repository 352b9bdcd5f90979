#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): tail threshold with the team front, small lone calls.
TAG=${1:-r3r}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
show() {
  python - $1 $2 <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); t = d['roofline']['per_kernel']['k_tail + k_front (tail rounds)']; b = d['roofline']['per_kernel']['k_optimize<1> (host-visible rounds)']
    print('%-14s' % sys.argv[2], round(d['value'], 1), 'maps/s', d['config']['host_threads_per_gpu'], 'thr', 'ms/step', round(d['ms_per_step'], 2), 'bulk ms', round(b['avg_launch_ms'] * b['launches'] / d['steps'], 2), b['launches'], 'k_tail', round(t['k_tail_ms'] / d['steps'], 2), t.get('k_tail_launches'), 'k_front', round(t['k_front_ms'] / d['steps'], 2), 'rounds', t.get('k_front_rounds_slowest_view'), 'att', t.get('k_front_attempts'))
except Exception as e:
    print(sys.argv[2], 'failed', e)
PY
}
one() { L=$1; shift; env "$@" timeout -s KILL 120 python bench.py --no-cpu-baseline --no-one-call --streams 1 --steps-per-call 1 --steps 10 --warmup 2 2>/dev/null > $OUT/b1_$L.json; show $OUT/b1_$L.json one_$L; }
one th12288
one th6144 MI_DMRECON_TAIL_THRESHOLD=6144
one th20000 MI_DMRECON_TAIL_THRESHOLD=20000
one th32768 MI_DMRECON_TAIL_THRESHOLD=32768
one th65536 MI_DMRECON_TAIL_THRESHOLD=65536
one th100k MI_DMRECON_TAIL_THRESHOLD=100000
for N in 1 3 5 10; do
  timeout -s KILL 100 python tools/trace_c3.py C3 $N 2>&1 | grep -E "phase (seeds|phase)|wall" | sed 's/\[mi_dmrecon\] phase//' | tr '\n' ' ' | cut -c1-200 | sed "s/^/lone $N views, default rule: /"; echo
done
MI_DMRECON_TRACE=1 timeout -s KILL 300 python tools/app_c3_timing.py 2>&1 | cut -c1-200 | tail -60 > $OUT/app_trace.txt
grep -v "(view)" $OUT/app_trace.txt | tail -14
timeout -s KILL 300 python -m pytest tests/test_gpu_dropin_app.py -x -q -m gpu 2>&1 | tail -3
