#!/bin/bash
# Runs ON THE GPU BOX: round 6, the collection on the final tree -- GPU suite, smoke, then tools/collect_profiles.sh r6.
export TMPDIR=/tmp
O=gpurun_out/r6
mkdir -p $O
( timeout -s KILL 1100 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "rc $?" >> $O/pytest_gpu.log ); tail -4 $O/pytest_gpu.log
( timeout -s KILL 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc $?" >> $O/smoke.log ); tail -2 $O/smoke.log
bash tools/collect_profiles.sh r6
