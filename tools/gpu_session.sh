#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the whole GPU suite, then the profile collection of the round.
TAG=${1:-r3}
export TMPDIR=/tmp
timeout -s KILL 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/collect_profiles.sh $TAG
