#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): round 5, session Y -- the final build (the reference's seed semantics in the seed launch
# as the default, globalVSMax <= 128, nrReconNeighbors <= 16): the whole GPU suite, smoke(), then the round's collection for the C3
# bench (tools/collect_profiles.sh, short form: bench lines, kernel stats, PMC passes at both call plans, round trace).
export TMPDIR=/tmp
O=gpurun_out/r5y
mkdir -p $O
( timeout -s KILL 1100 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "rc $?" >> $O/pytest_gpu.log )
tail -8 $O/pytest_gpu.log
( timeout -s KILL 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc $?" >> $O/smoke.log ); tail -3 $O/smoke.log
SKIP_EXTRAS=1 bash tools/collect_profiles.sh r5 > $O/collect.log 2>&1
tail -5 $O/collect.log | cut -c1-400
