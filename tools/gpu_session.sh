#!/bin/bash
# Runs ON THE GPU BOX (through gpurun).  Round-2 measurement session.  Output: gpurun_out/$TAG/.
TAG=${1:-s14}
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
echo "== persistent tail + device view selection: equality tests first (short spin limit: a hang must not take the box)"
MI_DMRECON_TAIL_SPIN_MS=200 timeout -s KILL 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "persistent_tail or device_view" 2>&1 | tail -8
B1="python bench.py --steps 8 --warmup 2 --streams 1 --steps-per-call 1 --no-cpu-baseline"
BD="python bench.py --steps 30 --warmup 2 --no-cpu-baseline"
L0=$PWD/mve_amd/csrc/libmi_dmrecon.so
run1() { N=$1; shift
  echo "== $N: 1 stream"; env "$@" timeout -s KILL 240 $B1 > $OUT/b1_$N.json 2> $OUT/b1_$N.err; show $OUT/b1_$N.json; tail -1 $OUT/b1_$N.err | cut -c1-200
}
rund() { N=$1; shift
  echo "== $N: default"; env "$@" timeout -s KILL 300 $BD > $OUT/bd_$N.json 2> $OUT/bd_$N.err; show $OUT/bd_$N.json
}
run1 classic MI_DMRECON_TAIL_PERSIST=0
run1 team_auto MI_DMRECON_TAIL_SPIN_MS=500
run1 team_1024 MI_DMRECON_TAIL_SPIN_MS=500 MI_DMRECON_TAIL_PERSIST_MAX=1024
run1 team_2048 MI_DMRECON_TAIL_SPIN_MS=500 MI_DMRECON_TAIL_PERSIST_MAX=2048
run1 team_g256 MI_DMRECON_TAIL_SPIN_MS=500 MI_DMRECON_TAIL_PERSIST_GRID=256
run1 oneteam MI_DMRECON_TAIL_PERSIST=1 MI_DMRECON_TAIL_SPIN_MS=500
rund auto MI_DMRECON_TAIL_SPIN_MS=500
rund team_always MI_DMRECON_TAIL_PERSIST=2 MI_DMRECON_TAIL_SPIN_MS=500
rund team_always_g128 MI_DMRECON_TAIL_PERSIST=2 MI_DMRECON_TAIL_SPIN_MS=500 MI_DMRECON_TAIL_PERSIST_GRID=128
echo "== trace of one call (auto)"
MI_DMRECON_TRACE=1 timeout -s KILL 200 python bench.py --steps 1 --warmup 1 --streams 1 --steps-per-call 1 --no-cpu-baseline 2>&1 | grep -E "phase|launch " > $OUT/trace_team.txt; grep phase $OUT/trace_team.txt | tail -6; grep launch $OUT/trace_team.txt | tail -40 | head -34
echo "== pytest parity + fullsize"; MI_DMRECON_TAIL_SPIN_MS=500 timeout -s KILL 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q --maxfail=4 2>&1 | tail -6
