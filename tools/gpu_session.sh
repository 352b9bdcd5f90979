#!/bin/bash
# Runs ON THE GPU BOX (through gpurun).  The measurement / test session of the day; this one: the whole GPU suite and smoke.
TAG=${1:-s39}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout -s KILL 1200 python -m pytest tests -m gpu -q --maxfail=5 2>&1 | tail -5
echo "== smoke"; timeout -s KILL 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
