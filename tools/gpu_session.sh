#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r4u
mkdir -p $OUT
timeout -s KILL 900 python -m pytest tests -q -m gpu 2>&1 | tail -150 > $OUT/pytest.txt; grep -E "^FAILED|^ERROR|passed|failed" $OUT/pytest.txt
timeout -s KILL 300 python tools/big_batch_probe.py 20 2>&1 | tail -6 | cut -c1-110
for I in 1 2 3; do
timeout -s KILL 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --one-call-n 30 2>$OUT/bench$I.err > $OUT/bench$I.json
python - $OUT/bench$I.json <<'PY'
import sys, json
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
oc = d["one_call"]
print("driver value %.1f" % d["value"], "repeats", [round(x) for x in d["repeats"]], "one_call %.2f ms (min %.2f) bulk %.2f front %.2f" % (oc["ms_per_call"], oc["ms_per_call_min_max"][0], oc["ms_bulk_kernel"], oc["ms_front_kernel"]))
PY
done
