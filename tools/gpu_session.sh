#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the driver's command, five times, front wall clocks per region.
export TMPDIR=/tmp
OUT=gpurun_out/r4b
mkdir -p $OUT
for k in 1 2 3 4 5; do
  MI_BENCH_REGION_LOG=1 timeout -s KILL 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --one-call-n 10 > $OUT/q.json 2> $OUT/q.err
  python - <<PY
import json,re
d=json.loads(open("$OUT/q.json").read().strip().splitlines()[-1])
fr=[float(m) for m in re.findall(r"front ([0-9.]+), download", open("$OUT/q.err").read())]
print("run $k", round(d["value"],1), [round(x) for x in d["repeats"]], "front wall", fr, "one_call", round(d["one_call"]["ms_per_call"],2))
PY
done
