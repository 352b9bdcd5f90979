#!/bin/bash
# Runs ON THE GPU BOX (through gpurun).  One round-2 measurement session: GPU tests, shader-clock stamps of one
# patch, then A/B bench lines of the texel-window variants (MI_DMRECON_WIN: 0 = gathers only, 1 = windows in the
# latency layout / tail, 3 = both layouts), round traces and rocprofv3 kernel stats.  Output: gpurun_out/$TAG/.
TAG=${1:-s2}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
show() { python - "$1" <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d['roofline']
    print('  value %.1f maps/s  ms/step %.2f  frac %.4f' % (d['value'], d['ms_per_step'], r['frac']))
    n = max(d['steps'], 1)
    print('  per step: bulk %.2f ms in %d launches, tail %.2f ms in %d launches; n_pass %d stages %d gather-passes %d'
          % (r['ms_bulk'] / n, r['n_bulk_launches'] / n, r['ms_tail'] / n, r['n_tail_launches'] / n, r['n_pass'] / n,
             r.get('n_window_stages', 0) / n, r.get('n_gather_passes', 0) / n))
except Exception as e:
    print('  (no json)', e)
PY
}
echo "== pytest"; timeout -s KILL 900 python -m pytest tests -m gpu -q --maxfail=6 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $OUT/pytest.log | cut -c1-220
echo "== timing probe"; timeout -s KILL 300 python tools/timing_probe.py > $OUT/timing_probe.txt 2>&1; cat $OUT/timing_probe.txt | cut -c1-1500
B1="python bench.py --steps 6 --warmup 2 --streams 1 --steps-per-call 1 --no-cpu-baseline"
BD="python bench.py --steps 30 --warmup 2 --no-cpu-baseline"
for W in 0 1 3; do
  echo "== bench 1 stream WIN=$W"
  MI_DMRECON_WIN=$W timeout -s KILL 240 $B1 > $OUT/bench1_win$W.json 2> $OUT/bench1_win$W.err; show $OUT/bench1_win$W.json; tail -3 $OUT/bench1_win$W.err
done
echo "== bench 1 stream WIN=3, 12x12 windows"
MI_DMRECON_LIB=$PWD/build/libmi_dmrecon_w12.so MI_DMRECON_WIN=3 timeout -s KILL 240 $B1 > $OUT/bench1_w12.json 2> $OUT/bench1_w12.err; show $OUT/bench1_w12.json; tail -3 $OUT/bench1_w12.err
for T in 2048 4096; do
  echo "== bench 1 stream WIN=1 tail threshold $T"
  MI_DMRECON_TAIL_THRESHOLD=$T MI_DMRECON_WIN=1 timeout -s KILL 240 $B1 > $OUT/bench1_win1_t$T.json 2> $OUT/bench1_win1_t$T.err; show $OUT/bench1_win1_t$T.json
done
echo "== bench default WIN=1"
MI_DMRECON_WIN=1 timeout -s KILL 300 $BD > $OUT/benchd_win1.json 2> $OUT/benchd_win1.err; show $OUT/benchd_win1.json
echo "== trace WIN=1"
MI_DMRECON_WIN=1 MI_DMRECON_TRACE=1 timeout -s KILL 240 python bench.py --steps 1 --warmup 1 --streams 1 --steps-per-call 1 --no-cpu-baseline > /dev/null 2> $OUT/trace_win1.txt
grep "phase" $OUT/trace_win1.txt | tail -6; grep "optimise launch" $OUT/trace_win1.txt | tail -n +620 | awk 'NR%40==1'
echo "== rocprofv3 kernel stats WIN=1"
MI_DMRECON_WIN=1 timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_win1 -o bench -- $B1 > $OUT/ks_win1.log 2>&1
find $OUT/ks_win1 -name "*kernel_stats.csv" | head -1 | xargs -r head -8
find $OUT/ks_win1 -name "*_kernel_trace.csv" -delete
du -sh $OUT
