#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): round 5, session I -- an iteration cap in the FAST kernel (patches still at work at
# iteration 7 / 10 are abandoned and redone by the general kernel), same lease.
export TMPDIR=/tmp
O=gpurun_out/r5i
mkdir -p $O
line() { python - "$1" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
oc = j.get("one_call") or {}
r = j["roofline"]
print("%s: value %.1f %s | bulk frac %.3f n_pass/n_eval %.4f | one_call %.2f ms (bulk %.2f)" % (sys.argv[1], j["value"], [round(v) for v in j["repeats"]],
      r["bulk_kernel_frac"], r["n_pass"] / r["n_eval"], oc.get("ms_per_call", 0), oc.get("ms_bulk_kernel", 0)))
PY
}
AB="--steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --distinct-scenes 0 --one-call-n 20"
for V in main cap7 cap10 main2; do
  L=mve_amd/csrc/libmi_dmrecon.so; case $V in cap*) L=build/libmi_dmrecon_$V.so;; esac
  MI_BENCH_REGION_LOG=1 MI_DMRECON_LIB=$PWD/$L timeout -s KILL 300 python bench.py $AB > $O/bench_$V.json 2> $O/bench_$V.err
  line $O/bench_$V.json; grep "^region" $O/bench_$V.err | tail -1
done
MI_DMRECON_LIB=$PWD/build/libmi_dmrecon_cap7.so timeout -s KILL 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "batch or front_kernel or reference_vectors" 2>&1 | tail -2
