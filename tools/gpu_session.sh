#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): round 5, session U -- the one-workgroup-per-view front with TWELVE wavefronts per
# workgroup at 168 registers (264 B of scratch per lane) instead of eight at 256: more attempts of a view in flight per CU.
export TMPDIR=/tmp
O=gpurun_out/r5u
mkdir -p $O
line() { python - "$1" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%s: value %.1f %s | bulk frac %.3f" % (sys.argv[1], j["value"], [round(v) for v in j["repeats"]], j["roofline"]["bulk_kernel_frac"]))
PY
}
AB="--steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --distinct-scenes 0 --no-one-call"
run() { V=$1; shift; env "$@" MI_BENCH_REGION_LOG=1 timeout -s KILL 240 python bench.py $AB > $O/bench_$V.json 2> $O/bench_$V.err; line $O/bench_$V.json; grep "^region" $O/bench_$V.err | sed -n '3p'; }
run fw12 MI_DMRECON_LIB=$PWD/build/libmi_dmrecon_fw12.so
run main A=1
