#!/bin/bash
# Runs ON THE GPU BOX (through gpurun).  Session r4d.
export TMPDIR=/tmp
OUT=gpurun_out/r4d
mkdir -p $OUT
timeout -s KILL 900 python -m pytest tests -q -m gpu 2>&1 | tail -150 > $OUT/pytest.txt; grep -E "^FAILED|^ERROR|passed|failed" $OUT/pytest.txt
timeout -s KILL 600 python tools/two_proc_probe.py > $OUT/two_proc.txt 2>&1; grep -v "^\[mi_dmrecon\]" $OUT/two_proc.txt | cut -c1-300
timeout -s KILL 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>$OUT/bench_driver.err > $OUT/bench_driver.json
python - $OUT/bench_driver.json <<'PY'
import sys, json
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
oc = d["one_call"]
print("value %.1f" % d["value"], "repeats", [round(x) for x in d["repeats"]], "one_call", {k: (round(v, 2) if isinstance(v, float) else v) for k, v in oc.items() if k != "what"})
PY
timeout -s KILL 300 python tools/lone_calls.py C3 10 > $OUT/lone_calls.json 2>$OUT/lone_calls.err; python -c "
import json; j=json.load(open('$OUT/lone_calls.json'))
for k,v in j['sizes'].items(): print(k, {a:(round(b,2) if isinstance(b,float) else b) for a,b in v.items()})"
