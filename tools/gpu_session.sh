#!/bin/bash
# Runs ON THE GPU BOX (through gpurun).  Final evidence session of round 2: all GPU tests, smoke, then the profile collection.
TAG=${1:-s35}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout -s KILL 1200 python -m pytest tests -m gpu -q --maxfail=5 2>&1 | tail -6
echo "== smoke"; timeout -s KILL 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "== profiles"; timeout -s KILL 1500 bash tools/collect_profiles.sh r2 2>&1 | tail -12
