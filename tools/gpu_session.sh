#!/bin/bash
# Runs ON THE GPU BOX (through gpurun).  One round-2 measurement session.  Output: gpurun_out/$TAG/.
TAG=${1:-s5}
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
    if 'strong_scaling' in d: print('   strong:', d['strong_scaling']['value'])
    if 'parity' in d: print('   parity:', {k: v for k, v in d['parity'].items() if k != 'against'})
    if 'cpu_baseline' in d: print('   cpu:', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
except Exception as e:
    print('  (no json)', e)
PY
}
echo "== pytest"; timeout -s KILL 900 python -m pytest tests -m gpu -q -s --maxfail=6 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "H1 |passed|failed|Error" $OUT/pytest.log | cut -c1-400 | tail -20
echo "== round-1 tree (commit 20247e6) on this box: default bench, then 1 stream"
(cd build/r1tree && timeout -s KILL 300 python bench.py --steps 30 --warmup 2 --no-cpu-baseline > $OUT/r1_benchd.json 2> $OUT/r1_benchd.err; show $OUT/r1_benchd.json
 timeout -s KILL 300 python bench.py --steps 6 --warmup 2 --streams 1 --steps-per-call 1 --no-cpu-baseline > $OUT/r1_bench1.json 2> $OUT/r1_bench1.err; show $OUT/r1_bench1.json)
B1="python bench.py --steps 6 --warmup 2 --streams 1 --steps-per-call 1 --no-cpu-baseline"
echo "== bench 1 stream (defaults)"; timeout -s KILL 240 $B1 > $OUT/bench1.json 2> $OUT/bench1.err; show $OUT/bench1.json; tail -3 $OUT/bench1.err
for S in 0 256 1024; do
  echo "== bench default (6 threads, 5 steps per call) SPECULATE=$S"
  MI_DMRECON_SPECULATE=$S timeout -s KILL 300 python bench.py --steps 30 --warmup 2 --no-cpu-baseline > $OUT/benchd_s$S.json 2> $OUT/benchd_s$S.err; show $OUT/benchd_s$S.json
done
for T in 4 8 12; do
  echo "== bench $T threads"
  timeout -s KILL 300 python bench.py --steps 60 --warmup 2 --streams $T --no-cpu-baseline > $OUT/benchd_t$T.json 2> $OUT/benchd_t$T.err; show $OUT/benchd_t$T.json
done
echo "== strong path, one rank"
MI_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29655 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout -s KILL 300 python bench.py --steps 10 --warmup 2 --scaling strong --no-cpu-baseline > $OUT/bench_strong1.json 2> $OUT/bench_strong1.err; show $OUT/bench_strong1.json
echo "== PMC calibration"; timeout -s KILL 600 bash tools/pmc_calib.sh > $OUT/pmc_calib.log 2>&1; tail -30 $OUT/pmc_calib.log | cut -c1-200
du -sh $OUT
