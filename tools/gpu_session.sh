#!/bin/bash
# Runs ON THE GPU BOX (through gpurun).  One round-2 measurement session.  Output: gpurun_out/$TAG/.
TAG=${1:-s4}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
show() { python - "$1" <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d['roofline']
    print('  value %.1f maps/s  ms/step %.2f  frac %.4f' % (d['value'], d['ms_per_step'], r['frac']))
    for k, v in r['per_kernel'].items():
        print('   %-40s launches/step %.0f avg %.4f ms  total/step %.2f ms frac %s' % (k, v['launches'] / d['steps'], v['avg_launch_ms'], v['launches'] * v['avg_launch_ms'] / d['steps'], v['frac']))
    if 'strong_scaling' in d: print('   strong:', d['strong_scaling']['value'])
    if 'parity' in d: print('   parity:', d['parity'])
    if 'cpu_baseline' in d: print('   cpu:', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
except Exception as e:
    print('  (no json)', e)
PY
}
echo "== pytest"; timeout -s KILL 900 python -m pytest tests -m gpu -q -s --maxfail=6 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "^H1|passed|failed|Error" $OUT/pytest.log | cut -c1-400 | tail -20
B1="python bench.py --steps 6 --warmup 2 --streams 1 --steps-per-call 1 --no-cpu-baseline"
for W in 0 1; do
  echo "== bench 1 stream WIN=$W"
  MI_DMRECON_WIN=$W timeout -s KILL 240 $B1 > $OUT/bench1_win$W.json 2> $OUT/bench1_win$W.err; show $OUT/bench1_win$W.json; tail -3 $OUT/bench1_win$W.err
done
echo "== bench 1 stream WIN=0 no speculation"
MI_DMRECON_SPECULATE=0 MI_DMRECON_WIN=0 timeout -s KILL 240 $B1 > $OUT/bench1_win0_nospec.json 2> $OUT/bench1_win0_nospec.err; show $OUT/bench1_win0_nospec.json
for W in 0 1; do
  for S in 0 1024; do
    echo "== bench default (6 threads, 5 steps per call) WIN=$W SPECULATE=$S"
    MI_DMRECON_SPECULATE=$S MI_DMRECON_WIN=$W timeout -s KILL 300 python bench.py --steps 30 --warmup 2 --no-cpu-baseline > $OUT/benchd_win${W}_s$S.json 2> $OUT/benchd_win${W}_s$S.err; show $OUT/benchd_win${W}_s$S.json
  done
done
echo "== trace WIN=0"
MI_DMRECON_WIN=0 MI_DMRECON_TRACE=1 timeout -s KILL 240 python bench.py --steps 2 --warmup 1 --streams 1 --steps-per-call 1 --no-cpu-baseline > /dev/null 2> $OUT/trace_win0.txt
grep "phase" $OUT/trace_win0.txt | tail -6
echo "== parity probe (fast reference)"; timeout -s KILL 400 python tools/parity_probe.py fast > $OUT/parity_probe_fast.txt 2>&1; cat $OUT/parity_probe_fast.txt | cut -c1-200
du -sh $OUT
