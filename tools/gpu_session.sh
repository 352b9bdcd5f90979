#!/bin/bash
# Runs ON THE GPU BOX (through gpurun).  Session r4b.
export TMPDIR=/tmp
OUT=gpurun_out/r4b
mkdir -p $OUT
timeout -s KILL 90 build/valu_rate2 > $OUT/valu_rate.txt 2>&1; grep -c cycles $OUT/valu_rate.txt; tail -3 $OUT/valu_rate.txt
timeout -s KILL 900 python -m pytest tests -q -m gpu 2>&1 | tail -150 > $OUT/pytest.txt; grep -E "^FAILED|^ERROR|passed|failed" $OUT/pytest.txt
timeout -s KILL 600 python tools/two_proc_probe.py > $OUT/two_proc.txt 2>&1; cat $OUT/two_proc.txt | cut -c1-400
