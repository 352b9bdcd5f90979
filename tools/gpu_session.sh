#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): round 5, session V -- two attempts per wavefront in the one-workgroup-per-view front
# (Lay<8, 4>, MI_DMRECON_FRONT_TWIN): bit-identity test, then the driver's plan with and without.
export TMPDIR=/tmp
O=gpurun_out/r5v
mkdir -p $O
true
line() { python - "$1" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%s: value %.1f %s | bulk frac %.3f" % (sys.argv[1], j["value"], [round(v) for v in j["repeats"]], j["roofline"]["bulk_kernel_frac"]))
PY
}
AB="--steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --distinct-scenes 0 --no-one-call"
run() { V=$1; shift; env "$@" MI_BENCH_REGION_LOG=1 timeout -s KILL 240 python bench.py $AB > $O/bench_$V.json 2> $O/bench_$V.err; line $O/bench_$V.json; grep "^region" $O/bench_$V.err | sed -n '3p'; }
run twin A=1
run single MI_DMRECON_FRONT_TWIN=0
run twin2 A=1
run single2 MI_DMRECON_FRONT_TWIN=0
