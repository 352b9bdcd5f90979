#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the round's profile collection + bench lines of configs 2 and 5.
export TMPDIR=/tmp
OUT=gpurun_out/r4
mkdir -p $OUT
bash tools/collect_profiles.sh r4 2>&1 | tail -12
timeout -s KILL 400 python bench.py --config C2 --steps 20 --warmup 3 --one-call-n 30 > $OUT/bench_c2.json 2> $OUT/bench_c2.err; tail -c 300 $OUT/bench_c2.json
timeout -s KILL 900 python bench.py --config C5 --steps 2 --warmup 1 --repeats 3 --one-call-n 10 > $OUT/bench_c5.json 2> $OUT/bench_c5.err; tail -c 300 $OUT/bench_c5.json; tail -3 $OUT/bench_c5.err
