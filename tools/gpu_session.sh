#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the measurement session of the day (tests + bench lines + probes in one call).
TAG=${1:-r3a}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -s KILL 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $OUT/pytest.log
cat $OUT/pytest.log
timeout -s KILL 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err
for F in 8 0 2 4 16 32 1000000; do
  MI_DMRECON_FRONT=$F timeout -s KILL 120 python bench.py --no-cpu-baseline --no-one-call --streams 1 --steps-per-call 1 --steps 10 --warmup 2 2>/dev/null > $OUT/b1_front$F.json
  python - $OUT/b1_front$F.json $F <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d['roofline']['per_kernel']
    t = r['k_tail + k_front (tail rounds)']
    print('front', sys.argv[2], ':', round(d['value'], 1), 'maps/s  ms/step', round(d['ms_per_step'], 2), ' bulk ms', round(r['k_optimize<1> (host-visible rounds)']['avg_launch_ms'] * r['k_optimize<1> (host-visible rounds)']['launches'] / d['steps'], 2),
          ' k_tail ms', round(t['k_tail_ms'] / d['steps'], 2), 'launches', t['k_tail_launches'] // d['steps'], ' k_front ms', round(t['k_front_ms'] / d['steps'], 2), 'rounds', t['k_front_rounds_slowest_view'] // d['steps'], 'attempts', t['k_front_attempts'] // d['steps'])
except Exception as e:
    print('front', sys.argv[2], 'failed', e)
PY
done
timeout -s KILL 120 python tools/trace_c3.py > $OUT/trace_c3.txt 2>&1
grep -E "phase|front view|total|wall" $OUT/trace_c3.txt | head -60
python - $OUT/bench_driver.json <<'PY'
import sys, json
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('driver line:', round(d['value'], 1), 'maps/s', d['config']['library_batches'], 'batches', 'frac', round(d['roofline']['frac'], 4), 'bulk frac', d['roofline']['bulk_kernel_frac'])
print('one_call:', json.dumps(d.get('one_call'))[:900])
print('parity:', json.dumps(d.get('parity'))[:600])
PY
