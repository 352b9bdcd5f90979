#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): round 5, session Z -- the one test that failed in session Y, then the round's collection
# for the C3 bench again with the profiled runs free of the seed-rule variant (tools/collect_profiles.sh, short form).
export TMPDIR=/tmp
O=gpurun_out/r5z
mkdir -p $O
( timeout -s KILL 600 python -m pytest tests -m gpu -q -k "sparse_maps or concurrent_calls" > $O/pytest_sel.log 2>&1; echo "rc $?" >> $O/pytest_sel.log )
tail -4 $O/pytest_sel.log
rm -rf gpurun_out/r5
SKIP_EXTRAS=1 bash tools/collect_profiles.sh r5 > $O/collect.log 2>&1
tail -4 $O/collect.log | cut -c1-300
