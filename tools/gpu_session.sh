#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): round 5, the last minutes -- the sixteen-slot layout at config 3's size (tools/wide_probe.py).
export TMPDIR=/tmp
mkdir -p gpurun_out/r5
timeout -s KILL 300 python tools/wide_probe.py > gpurun_out/r5/wide_probe.txt 2>&1; cat gpurun_out/r5/wide_probe.txt | tail -15
