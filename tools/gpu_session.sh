#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r4p
mkdir -p $OUT
for W in 150 1000 3000; do
MI_DMRECON_MERGE_WINDOW_US=$W timeout -s KILL 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-one-call 2>$OUT/bench$W.err > $OUT/bench$W.json
python - $OUT/bench$W.json $W <<'PY'
import sys, json
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("window", sys.argv[2], "value %.1f" % d["value"], "repeats", [round(x) for x in d["repeats"]], d["config"]["library_batch_log"])
PY
done
