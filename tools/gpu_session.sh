#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): round 5, session W -- globalVSMax up to 128 (two-word availability mask, the view
# selection's NCC table in dynamic shared memory) and nrReconNeighbors up to 16 (Lay<1, 16>, host-visible rounds only):
# the whole GPU suite (with the new scene W2), then the driver's plan on this build and on the build before (A/B, same lease).
export TMPDIR=/tmp
O=gpurun_out/r5w
mkdir -p $O
( timeout -s KILL 1100 python -m pytest tests -m gpu -x -q -s -k "wider_view_sets or wide_view_sets" > $O/pytest_wide.log 2>&1; echo "rc $?" >> $O/pytest_wide.log )
tail -5 $O/pytest_wide.log
grep "^W2\|^W1" $O/pytest_wide.log
( timeout -s KILL 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "rc $?" >> $O/pytest_gpu.log )
tail -6 $O/pytest_gpu.log
line() { python - "$1" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%s: value %.1f %s | bulk frac %.3f" % (sys.argv[1], j["value"], [round(v) for v in j["repeats"]], j["roofline"]["bulk_kernel_frac"]))
PY
}
AB="--steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --distinct-scenes 0 --no-one-call"
run() { V=$1; shift; env "$@" MI_BENCH_REGION_LOG=1 timeout -s KILL 240 python bench.py $AB > $O/bench_$V.json 2> $O/bench_$V.err; line $O/bench_$V.json; grep "^region" $O/bench_$V.err | sed -n '3p'; }
run new A=1
run base MI_DMRECON_LIB=$PWD/build/libmi_dmrecon_base.so
run new2 A=1
run base2 MI_DMRECON_LIB=$PWD/build/libmi_dmrecon_base.so
