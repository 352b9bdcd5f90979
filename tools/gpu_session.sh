#!/bin/bash
# Runs ON THE GPU BOX (through gpurun).  The GPU test suite under every opt-in mode of the library.
TAG=${1:-s26}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for M in "MI_DMRECON_TAIL_PERSIST=2 MI_DMRECON_TAIL_SPIN_MS=500" "MI_DMRECON_TAIL_PERSIST=1 MI_DMRECON_TAIL_SPIN_MS=500" "MI_DMRECON_GVS_DEVICE=1" "MI_DMRECON_BULK_TOKEN=1 MI_DMRECON_TAIL_PRIORITY=1" "MI_DMRECON_RESERVE_CUS=32" "MI_DMRECON_GVS_TABLES=0"; do
  echo "== $M"; env $M timeout -s KILL 900 python -m pytest tests -m gpu -q --maxfail=5 2>&1 | tail -4
done
