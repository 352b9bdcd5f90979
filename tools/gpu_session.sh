#!/bin/bash
# Runs ON THE GPU BOX (through gpurun).  Session r4g: packed speculative rounds, threshold sweep.
export TMPDIR=/tmp
OUT=gpurun_out/r4g
mkdir -p $OUT
timeout -s KILL 900 python -m pytest tests -q -m gpu 2>&1 | tail -150 > $OUT/pytest.txt; grep -E "^FAILED|^ERROR|passed|failed" $OUT/pytest.txt
for SP in 0 32768 150000 400000 1500000; do
MI_DMRECON_SPEC_ROUNDS=$SP timeout -s KILL 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --one-call-n 30 2>$OUT/bench_sp$SP.err > $OUT/bench_sp$SP.json
python - $OUT/bench_sp$SP.json $SP <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    oc = d["one_call"]
    print("spec", sys.argv[2], "value %.1f" % d["value"], "repeats", [round(x) for x in d["repeats"]], "one_call %.2f ms bulk %.2f front %.2f" % (oc["ms_per_call"], oc["ms_bulk_kernel"], oc["ms_front_kernel"]))
except Exception as e:
    print("spec", sys.argv[2], "FAILED", e, open(sys.argv[1].replace(".json", ".err")).read()[-800:])
PY
done
MI_DMRECON_SPEC_ROUNDS=400000 timeout -s KILL 120 python tools/trace_c3.py > $OUT/round_trace_c3.txt 2>&1; grep -E "phase|total|optimise launch" $OUT/round_trace_c3.txt | head -40
