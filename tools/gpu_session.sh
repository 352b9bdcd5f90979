#!/bin/bash
# Runs ON THE GPU BOX (through gpurun).  Session r4f: speculative small rounds.
export TMPDIR=/tmp
OUT=gpurun_out/r4f
mkdir -p $OUT
timeout -s KILL 900 python -m pytest tests -q -m gpu 2>&1 | tail -150 > $OUT/pytest.txt; grep -E "^FAILED|^ERROR|passed|failed" $OUT/pytest.txt
for SP in 0 32768 65536; do
MI_DMRECON_SPEC_ROUNDS=$SP timeout -s KILL 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --one-call-n 30 2>$OUT/bench_sp$SP.err > $OUT/bench_sp$SP.json
python - $OUT/bench_sp$SP.json $SP <<'PY'
import sys, json
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
oc = d["one_call"]
print("spec", sys.argv[2], "value %.1f" % d["value"], "repeats", [round(x) for x in d["repeats"]], "one_call %.2f ms bulk %.2f front %.2f" % (oc["ms_per_call"], oc["ms_bulk_kernel"], oc["ms_front_kernel"]))
PY
done
timeout -s KILL 120 python tools/trace_c3.py > $OUT/round_trace_c3.txt 2>&1; grep -E "phase|total|optimise launch" $OUT/round_trace_c3.txt | head -60
timeout -s KILL 300 python tools/lone_calls.py C3 8 > $OUT/lone_calls.json 2>$OUT/lone_calls.err; python -c "
import json; j=json.load(open('$OUT/lone_calls.json'))
for k,v in j['sizes'].items(): print(k, 'ms %.2f bulk %.2f front %.2f' % (v['ms_median'], v['ms_bulk_kernel'], v['ms_front_kernel']))"
