#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): round 5, session F -- finished views written into page-locked buffers by the device
# (one dispatch per view instead of a flatten launch and three copies): its test, bench lines with and without.
export TMPDIR=/tmp
O=gpurun_out/r5f
mkdir -p $O
line() { python - "$1" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
oc = j.get("one_call") or {}
print("%s: value %.1f %s | bulk frac %.3f | one_call %.2f ms (bulk %.2f front %.2f)" % (sys.argv[1], j["value"], [round(v) for v in j["repeats"]],
      j["roofline"]["bulk_kernel_frac"], oc.get("ms_per_call", 0), oc.get("ms_bulk_kernel", 0), oc.get("ms_front_kernel", 0)))
PY
}
timeout -s KILL 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "page_locked or front_kernel or batch" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
AB="--steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --distinct-scenes 0 --one-call-n 20"
MI_BENCH_REGION_LOG=1 timeout -s KILL 300 python bench.py $AB > $O/bench_emit.json 2> $O/bench_emit.err
line $O/bench_emit.json; grep "^region" $O/bench_emit.err | tail -2
MI_DMRECON_EMIT=0 MI_BENCH_REGION_LOG=1 timeout -s KILL 300 python bench.py $AB > $O/bench_copies.json 2> $O/bench_copies.err
line $O/bench_copies.json; grep "^region" $O/bench_copies.err | tail -1
MI_DMRECON_FRONT_ORDER=0 MI_BENCH_REGION_LOG=1 timeout -s KILL 300 python bench.py $AB --no-one-call > $O/bench_emit_order0.json 2> $O/bench_emit_order0.err
line $O/bench_emit_order0.json; grep "^region" $O/bench_emit_order0.err | tail -1
MI_DMRECON_TRACE=1 timeout -s KILL 100 python tools/trace_c3.py 2>&1 | grep -E "streamed back|phase C|download" | head -8
