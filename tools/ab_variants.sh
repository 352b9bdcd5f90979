#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): lone 20-view calls of C3 with variant builds of the library (MI_DMRECON_LIB).
#   bash tools/ab_variants.sh <name> [<name> ...]      name = "main" or the NAME of `make -C mve_amd/csrc variant`
export TMPDIR=/tmp
OUT=gpurun_out/ab
mkdir -p $OUT
for v in "$@"; do
  if [ $v = main ]; then unset MI_DMRECON_LIB; else export MI_DMRECON_LIB=$PWD/build/libmi_dmrecon_$v.so; fi
  timeout -s KILL 300 python bench.py --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline --one-call-n 30 > $OUT/$v.json 2> $OUT/$v.err || tail -3 $OUT/$v.err
  python - <<PY
import json
d=json.loads(open("$OUT/$v.json").read().strip().splitlines()[-1])
o=d["one_call"]
print("%-8s one_call %.2f ms (min %.2f)  bulk %.2f  front %.2f  plan %.2f | plan value %.0f" % ("$v", o["ms_per_call"], o["ms_per_call_min_max"][0], o["ms_bulk_kernel"], o["ms_front_kernel"], o["ms_host_planning"], d["value"]))
PY
done
