#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the driver's command on the final binary (region log) + the parity tests.
export TMPDIR=/tmp
OUT=gpurun_out/r4
mkdir -p $OUT
MI_BENCH_REGION_LOG=1 timeout -s KILL 70 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err
python - <<PY
import json
d=json.loads(open("$OUT/bench_driver.json").read().strip().splitlines()[-1])
print(round(d["value"],1), [round(x) for x in d["repeats"]], "one_call", round(d["one_call"]["ms_per_call"],2), d["parity"]["within_bounds"])
PY
grep "^region" $OUT/bench_driver.err | cut -c1-170 | head -3
timeout -s KILL 40 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -1
