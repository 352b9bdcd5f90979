#!/bin/bash
# Runs ON THE GPU BOX (through gpurun).  Round-2 evidence session: GPU tests, the profile set (bench lines, rocprofv3
# kernel stats, PMC passes), BASELINE config 5 at full size.  Output: gpurun_out/$TAG/, gpurun_out/r2/.
TAG=${1:-s6}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest"; timeout -s KILL 900 python -m pytest tests -m gpu -q --maxfail=6 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log | cut -c1-300
echo "== profiles"; timeout -s KILL 1500 bash tools/collect_profiles.sh r2 > $OUT/collect.log 2>&1; tail -5 $OUT/collect.log | cut -c1-700
echo "== C5 full size"; timeout -s KILL 900 python tools/c5_full.py > $OUT/c5_full.json 2> $OUT/c5_full.err; tail -c 2500 $OUT/c5_full.json; tail -3 $OUT/c5_full.err
du -sh $OUT gpurun_out/r2
