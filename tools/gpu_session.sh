#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): where a latency-layout attempt of the front kernel spends its cycles (MI_PROBE build).
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout -s KILL 25 python tools/patch_probe.py > gpurun_out/r4/patch_probe.txt 2>&1
tail -32 gpurun_out/r4/patch_probe.txt
