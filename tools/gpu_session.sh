#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): round 5, session W -- globalVSMax up to 128 (two-word availability mask, the view
# selection's NCC table in dynamic shared memory) and nrReconNeighbors up to 16 (Lay<1, 16>, host-visible rounds only):
# the whole GPU suite (with the new scene W2), then the driver's plan on this build and on the build before (A/B, same lease).
export TMPDIR=/tmp
O=gpurun_out/r5w
mkdir -p $O
timeout -s KILL 300 python tools/w2_probe.py > $O/w2_probe.txt 2>&1; grep "SEED_REOPT\|^k\|only HIP\|PatchOpt" $O/w2_probe.txt
( timeout -s KILL 1100 python -m pytest tests -m gpu -q -s -k "wider_view_sets or wide_view_sets" > $O/pytest_wide.log 2>&1; echo "rc $?" >> $O/pytest_wide.log )
tail -5 $O/pytest_wide.log
grep "^W2\|^W1\|Error\|assert" $O/pytest_wide.log | head -30
( timeout -s KILL 1100 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "rc $?" >> $O/pytest_gpu.log )
tail -8 $O/pytest_gpu.log
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
# the setup phase of a 400-view batch on both builds (MI_DMRECON_TRACE: the leader of a merged batch prints its phases)
for V in new base; do
  L=$PWD/mve_amd/csrc/libmi_dmrecon.so; [ $V = base ] && L=$PWD/build/libmi_dmrecon_base.so
  MI_DMRECON_LIB=$L MI_DMRECON_TRACE=1 timeout -s KILL 200 python bench.py --steps 20 --warmup 2 --repeats 1 --no-cpu-baseline --distinct-scenes 0 --no-one-call > /dev/null 2> $O/trace_$V.err
  echo "== $V"; grep "upload:\|setup + uploads" $O/trace_$V.err | tail -8
done
