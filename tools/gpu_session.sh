#!/bin/bash
# Runs ON THE GPU BOX (through gpurun).  Round-2 evidence session: GPU tests, the profile set (bench lines, rocprofv3
# kernel stats, PMC passes), single-patch timing probe, round trace.  Output: gpurun_out/$TAG/, gpurun_out/r2/.
TAG=${1:-s10}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest"; timeout -s KILL 900 python -m pytest tests -m gpu -q --maxfail=6 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log | cut -c1-300
echo "== smoke"; timeout -s KILL 300 python __graft_entry__.py smoke 2>&1 | tail -2
echo "== profiles"; timeout -s KILL 1500 bash tools/collect_profiles.sh r2 > $OUT/collect.log 2>&1; tail -3 $OUT/collect.log | cut -c1-400
echo "== timing probe"; timeout -s KILL 300 python tools/timing_probe.py > $OUT/timing_probe.txt 2>&1; grep -E "^lpv|^total|^by closing" $OUT/timing_probe.txt | cut -c1-400
echo "== round trace"; MI_DMRECON_TRACE=1 timeout -s KILL 240 python bench.py --steps 2 --warmup 1 --streams 1 --steps-per-call 1 --no-cpu-baseline > /dev/null 2> $OUT/trace.txt; grep "phase" $OUT/trace.txt | tail -6
echo "== drop-in app on the C3 scene"; timeout -s KILL 600 python tools/app_c3_timing.py 2>&1 | tail -4
du -sh $OUT gpurun_out/r2
