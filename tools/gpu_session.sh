#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the bench line of config 5 on the round's final binary.
export TMPDIR=/tmp
OUT=gpurun_out/r4
mkdir -p $OUT
timeout -s KILL 235 python bench.py --config C5 --steps 2 --warmup 1 --repeats 3 --one-call-n 10 > $OUT/bench_c5.json 2> $OUT/bench_c5.err; tail -c 600 $OUT/bench_c5.json; tail -2 $OUT/bench_c5.err
