#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): round 5, last session -- the whole GPU suite once more on the final tree (a second
# pass: the timing-dependent tests), then lone calls of a rank's share (the strong-scaling prediction's input).
export TMPDIR=/tmp
O=gpurun_out/r5f
mkdir -p $O gpurun_out/r5
( timeout -s KILL 1100 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "rc $?" >> $O/pytest_gpu.log )
tail -6 $O/pytest_gpu.log
timeout -s KILL 330 python tools/lone_calls.py C3 12 > gpurun_out/r5/lone_calls.json 2> gpurun_out/r5/lone_calls.err
cut -c1-400 gpurun_out/r5/lone_calls.json
