#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the round's final profile collection + the bench line of config 2 + the drop-in tests.
export TMPDIR=/tmp
OUT=gpurun_out/r4
mkdir -p $OUT
bash tools/collect_profiles.sh r4 2>&1 | tail -6
timeout -s KILL 400 python bench.py --config C2 --steps 20 --warmup 3 --one-call-n 30 > $OUT/bench_c2.json 2> $OUT/bench_c2.err; tail -c 200 $OUT/bench_c2.json
timeout -s KILL 600 python -m pytest tests/test_gpu_dropin_app.py -x -q 2>&1 | tail -2
