#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): round 5, session N -- footprint records block-linear (8 x 8 per tile, -DMI_TILED_QUADS)
# against the row-major records, one scene and 20 distinct scenes, with the L2 counters of both.
export TMPDIR=/tmp
O=gpurun_out/r5n
mkdir -p $O
line() { python - "$1" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
oc = j.get("one_call") or {}
d = j["config"].get("distinct_scenes_variant") or {}
print("%s: value %.1f %s | bulk frac %.3f | one_call %.2f ms (bulk %.2f) | distinct %.1f (bulk %.2f ms/step)" % (
      sys.argv[1], j["value"], [round(v) for v in j["repeats"]], j["roofline"]["bulk_kernel_frac"], oc.get("ms_per_call", 0),
      oc.get("ms_bulk_kernel", 0), d.get("value", 0), d.get("ms_bulk_kernel_per_step", 0)))
PY
}
MI_DMRECON_LIB=$PWD/build/libmi_dmrecon_tiled.so timeout -s KILL 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "reference_vectors or maps_vs_reference or batch" 2>&1 | tail -2
AB="--steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --distinct-scenes 20 --one-call-n 20"
for V in base tiled base2 tiled2; do
  L=mve_amd/csrc/libmi_dmrecon.so; case $V in tiled*) L=build/libmi_dmrecon_tiled.so;; esac
  MI_BENCH_REGION_LOG=1 MI_DMRECON_LIB=$PWD/$L timeout -s KILL 300 python bench.py $AB > $O/bench_$V.json 2> $O/bench_$V.err
  line $O/bench_$V.json; grep "^region" $O/bench_$V.err | sed -n '3p'
done
R=$PWD; cd /tmp
for V in base tiled; do
  L=$R/mve_amd/csrc/libmi_dmrecon.so; case $V in tiled*) L=$R/build/libmi_dmrecon_tiled.so;; esac
  for C in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE"; do
    D=$R/$O/pmc_${V}_$(echo $C | tr ' ' '+')
    MI_DMRECON_LIB=$L timeout -s KILL 150 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D -o bench -- python $R/bench.py --steps 20 --warmup 1 --repeats 1 --no-cpu-baseline --no-one-call --distinct-scenes 0 > $D.log 2>&1
    python - "$D" "$V" <<'PY'
import csv, collections, glob, os, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        if "k_optimize<mi_fw5::Lay<1, 4>, true" in k or "k_optimize<mi_fw5::Lay<1, 4>, false, true" in k:
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
for k, c in acc.items():
    print(sys.argv[2], k[:60], {n: "%.4g" % v for n, v in c.items()})
PY
  done
done
cd $R; find $O -name "*_kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
