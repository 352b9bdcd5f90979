#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r4b
mkdir -p $OUT
timeout -s KILL 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -3
MI_BENCH_REGION_LOG=1 timeout -s KILL 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --one-call-n 20 > $OUT/q.json 2> $OUT/q.err
python - <<PY
import json,re
d=json.loads(open("$OUT/q.json").read().strip().splitlines()[-1])
print(round(d["value"],1), [round(x) for x in d["repeats"]], "one_call", round(d["one_call"]["ms_per_call"],2))
PY
grep "^region" $OUT/q.err | cut -c1-200
