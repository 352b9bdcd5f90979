#!/bin/bash
# Runs ON THE GPU BOX (through gpurun).  Round-2 measurement session.  Output: gpurun_out/$TAG/.
TAG=${1:-s9}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
show() { python - "$1" <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d['roofline']
    n = d['steps']
    line = '  value %.1f maps/s  ms/step %.2f  frac %.4f' % (d['value'], d['ms_per_step'], r['frac'])
    if 'per_kernel' in r:
        line += ' | ' + ' | '.join('%s: %.0f x %.4f = %.2f ms' % (k[:10], v['launches']/n, v['avg_launch_ms'], v['launches']*v['avg_launch_ms']/n) for k, v in r['per_kernel'].items())
    print(line)
except Exception as e:
    print('  (no json)', e)
PY
}
echo "== pytest"; timeout -s KILL 900 python -m pytest tests -m gpu -q -s --maxfail=8 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "H1 |filter width|passed|failed|Error" $OUT/pytest.log | cut -c1-300 | tail -14
B1="python bench.py --steps 8 --warmup 2 --streams 1 --steps-per-call 1 --no-cpu-baseline"
echo "== 1 stream"; timeout -s KILL 240 $B1 > $OUT/b1.json 2> $OUT/b1.err; show $OUT/b1.json; tail -2 $OUT/b1.err
echo "== default"; timeout -s KILL 400 python bench.py --steps 30 --warmup 2 --no-cpu-baseline > $OUT/bd.json 2> $OUT/bd.err; show $OUT/bd.json
