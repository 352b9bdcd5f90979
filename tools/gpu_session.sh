#!/bin/bash
# Runs ON THE GPU BOX (through gpurun).  One round-2 measurement session: GPU tests, then bench lines
# (MI_DMRECON_WIN: 0 = gathers only, 1 = texel windows in the latency layout / tail, 3 = both layouts),
# a round trace and rocprofv3 kernel stats.  Output: gpurun_out/$TAG/.
TAG=${1:-s3}
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
echo "== pytest"; timeout -s KILL 900 python -m pytest tests -m gpu -q --maxfail=6 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $OUT/pytest.log | cut -c1-220
B1="python bench.py --steps 6 --warmup 2 --streams 1 --steps-per-call 1 --no-cpu-baseline"
for W in 0 1; do
  echo "== bench 1 stream WIN=$W"
  MI_DMRECON_WIN=$W timeout -s KILL 240 $B1 > $OUT/bench1_win$W.json 2> $OUT/bench1_win$W.err; show $OUT/bench1_win$W.json; tail -3 $OUT/bench1_win$W.err
done
for T in 4096 30000; do
  echo "== bench 1 stream WIN=1 tail threshold $T"
  MI_DMRECON_TAIL_THRESHOLD=$T MI_DMRECON_WIN=1 timeout -s KILL 240 $B1 > $OUT/bench1_win1_t$T.json 2> $OUT/bench1_win1_t$T.err; show $OUT/bench1_win1_t$T.json
done
echo "== bench default (6 threads, 5 steps per call) WIN=1, with cpu baseline + parity"
MI_DMRECON_WIN=1 timeout -s KILL 400 python bench.py --steps 30 --warmup 2 > $OUT/benchd_win1.json 2> $OUT/benchd_win1.err; show $OUT/benchd_win1.json; tail -3 $OUT/benchd_win1.err
echo "== bench default WIN=0"
MI_DMRECON_WIN=0 timeout -s KILL 300 python bench.py --steps 30 --warmup 2 --no-cpu-baseline > $OUT/benchd_win0.json 2> $OUT/benchd_win0.err; show $OUT/benchd_win0.json
echo "== strong-scaling path with one rank (MI_FORCE_DIST)"
MI_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29655 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MI_DMRECON_WIN=1 timeout -s KILL 300 python bench.py --steps 10 --warmup 2 --scaling strong --no-cpu-baseline > $OUT/bench_strong1.json 2> $OUT/bench_strong1.err; show $OUT/bench_strong1.json; tail -2 $OUT/bench_strong1.err
echo "== trace WIN=1"
MI_DMRECON_WIN=1 MI_DMRECON_TRACE=1 timeout -s KILL 240 python bench.py --steps 1 --warmup 1 --streams 1 --steps-per-call 1 --no-cpu-baseline > /dev/null 2> $OUT/trace_win1.txt
grep "phase" $OUT/trace_win1.txt | tail -6; grep "optimise launch" $OUT/trace_win1.txt | tail -n +620 | awk 'NR%40==1'
echo "== rocprofv3 kernel stats WIN=1"
MI_DMRECON_WIN=1 timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_win1 -o bench -- $B1 > $OUT/ks_win1.log 2>&1
find $OUT/ks_win1 -name "*kernel_stats.csv" | head -1 | xargs -r head -8
find $OUT/ks_win1 -name "*_kernel_trace.csv" -delete
du -sh $OUT
