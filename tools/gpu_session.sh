#!/bin/bash
# Runs ON THE GPU BOX (through gpurun).  Session r4c.
export TMPDIR=/tmp
OUT=gpurun_out/r4c
mkdir -p $OUT
timeout -s KILL 90 build/valu_rate2 > $OUT/valu_rate.txt 2>&1; grep cycles $OUT/valu_rate.txt | tail -34
timeout -s KILL 900 python -m pytest tests -q -m gpu 2>&1 | tail -150 > $OUT/pytest.txt; grep -E "^FAILED|^ERROR|passed|failed" $OUT/pytest.txt
timeout -s KILL 600 python tools/two_proc_probe.py > $OUT/two_proc.txt 2>&1; grep -v "^\[mi_dmrecon\]" $OUT/two_proc.txt | cut -c1-300
timeout -s KILL 300 python bench.py --gpus 1 --steps 20 --warmup 5 2>$OUT/bench_driver.err > $OUT/bench_driver.json; tail -c 2500 $OUT/bench_driver.json
