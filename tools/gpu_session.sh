#!/bin/bash
# Runs ON THE GPU BOX (through gpurun).  One round-2 measurement session: GPU tests, then A/B bench lines of the
# texel-window variants (MI_DMRECON_WIN: 0 = gathers only, 1 = windows in the latency layout / tail, 3 = both
# layouts), round traces and rocprofv3 kernel stats.  Everything lands in gpurun_out/$TAG/.
TAG=${1:-s1}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest"; timeout -s KILL 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest.log
B1="python bench.py --steps 6 --warmup 2 --streams 1 --steps-per-call 1 --no-cpu-baseline"
BD="python bench.py --steps 30 --warmup 2 --no-cpu-baseline"
for W in 0 1 3; do
  echo "== bench 1 stream WIN=$W"
  MI_DMRECON_WIN=$W timeout -s KILL 240 $B1 > $OUT/bench1_win$W.json 2> $OUT/bench1_win$W.err; tail -c 1500 $OUT/bench1_win$W.json | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
    print('  value %.1f maps/s  ms/step %.2f  frac %.4f  avg_launch_ms %.4f launches %d' % (d['value'], d['ms_per_step'], r['frac'], r['avg_launch_ms'], r['launches']), {k: r.get(k) for k in ('ms_bulk', 'ms_tail', 'n_bulk_launches', 'n_tail_launches')})
except Exception as e:
    print('  (no json)', e)
"
  tail -3 $OUT/bench1_win$W.err
done
for W in 1 3; do
  echo "== bench default WIN=$W"
  MI_DMRECON_WIN=$W timeout -s KILL 300 $BD > $OUT/benchd_win$W.json 2> $OUT/benchd_win$W.err; python -c "
import sys, json
try:
    d = json.loads(open('$OUT/benchd_win$W.json').read().strip().splitlines()[-1]); print('  value %.1f maps/s  ms/step %.2f' % (d['value'], d['ms_per_step']))
except Exception as e:
    print('  (no json)', e)
"
done
for W in 1 3; do
  echo "== trace WIN=$W"
  MI_DMRECON_WIN=$W MI_DMRECON_TRACE=1 timeout -s KILL 240 python bench.py --steps 1 --warmup 1 --streams 1 --steps-per-call 1 --no-cpu-baseline > /dev/null 2> $OUT/trace_win$W.txt
  grep "phase" $OUT/trace_win$W.txt | tail -7
done
for W in 1 3; do
  echo "== rocprofv3 kernel stats WIN=$W"
  MI_DMRECON_WIN=$W timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_win$W -o bench -- $B1 > $OUT/ks_win$W.log 2>&1
  find $OUT/ks_win$W -name "*kernel_stats.csv" | head -1 | xargs -r head -8
  find $OUT/ks_win$W -name "*_kernel_trace.csv" -delete
done
du -sh $OUT
