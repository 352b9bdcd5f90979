#!/bin/bash
# Runs ON THE GPU BOX (through gpurun).  Merged concurrent calls: tests, then A/B of the default bench.
TAG=${1:-s34}
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
        line += ' | ' + ' | '.join('%s: %.1f x %.4f = %.2f ms' % (k[:10], v['launches']/n, v['avg_launch_ms'], v['launches']*v['avg_launch_ms']/n) for k, v in r['per_kernel'].items())
    print(line)
except Exception as e:
    print('  (no json)', e)
PY
}
echo "== test"; timeout -s KILL 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "concurrent_calls" 2>&1 | tail -40
