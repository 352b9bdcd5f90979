#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the drop-in app after the shim's start-up phases were overlapped.
export TMPDIR=/tmp
mkdir -p gpurun_out/r3al
timeout -s KILL 300 python -m pytest tests/test_gpu_dropin_app.py -x -q -m gpu 2>&1 | tail -3
MI_DMRECON_TRACE=1 timeout -s KILL 300 python tools/app_c3_timing.py 2>&1 | grep -v "^\[mi_dmrecon\]" > gpurun_out/r3al/app_c3_timing.txt
grep -v "(view)" gpurun_out/r3al/app_c3_timing.txt | tail -16
