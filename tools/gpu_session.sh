#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): round 5, the closing pass on the final commit -- what the driver runs at the end of a
# round: the whole GPU suite and smoke() (the bench lines of this tree: profiles/r5_bench_*.json, tools/collect_profiles.sh).
export TMPDIR=/tmp
O=gpurun_out/r5end
mkdir -p $O
( timeout -s KILL 1100 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "rc $?" >> $O/pytest_gpu.log ); tail -4 $O/pytest_gpu.log
( timeout -s KILL 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc $?" >> $O/smoke.log ); tail -2 $O/smoke.log
