#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): round 5, session P -- two samples at a time in packed f32 (v_pk_fma_f32 ...):
# the product build (consume step packed) against the build before it (base) and the build with the geometry packed
# as well (pkg: -DMI_PK_GEOM); GPU test suite on the product build first.
export TMPDIR=/tmp
O=gpurun_out/r5p
mkdir -p $O
line() { python - "$1" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
oc = j.get("one_call") or {}
print("%s: value %.1f %s | bulk frac %.3f | one_call %.2f ms (bulk %.2f, front %.2f)" % (
      sys.argv[1], j["value"], [round(v) for v in j["repeats"]], j["roofline"]["bulk_kernel_frac"], oc.get("ms_per_call", 0),
      oc.get("ms_bulk_kernel", 0), oc.get("ms_front_kernel", 0)))
PY
}
timeout -s KILL 420 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
AB="--steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --distinct-scenes 0 --one-call-n 20"
for V in base main pkg base2 main2; do
  L=mve_amd/csrc/libmi_dmrecon.so
  case $V in base*) L=build/libmi_dmrecon_base.so;; pkg*) L=build/libmi_dmrecon_pkg.so;; esac
  MI_BENCH_REGION_LOG=1 MI_DMRECON_LIB=$PWD/$L timeout -s KILL 240 python bench.py $AB > $O/bench_$V.json 2> $O/bench_$V.err
  line $O/bench_$V.json; grep "^region" $O/bench_$V.err | sed -n '3p'
done
