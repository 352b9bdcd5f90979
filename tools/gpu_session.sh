#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): round 5, session H -- latency layout: cross-row sums by v_permlane swaps against the
# v_readlane form (-DMI_LAT_READLANE), same lease; the parity file on the new build.
export TMPDIR=/tmp
O=gpurun_out/r5h
mkdir -p $O
line() { python - "$1" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
oc = j.get("one_call") or {}
print("%s: value %.1f %s | one_call %.2f ms (bulk %.2f front %.2f) | merged front per step %.2f ms" % (sys.argv[1], j["value"], [round(v) for v in j["repeats"]],
      oc.get("ms_per_call", 0), oc.get("ms_bulk_kernel", 0), oc.get("ms_front_kernel", 0), j["roofline"]["per_kernel"]["k_tail + k_front (tail rounds)"]["k_front_ms"] / (j["steps"] * len(j["repeats"]))))
PY
}
timeout -s KILL 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
AB="--steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --distinct-scenes 0 --one-call-n 40"
for V in main rl main2 rl2; do
  L=mve_amd/csrc/libmi_dmrecon.so; case $V in rl*) L=build/libmi_dmrecon_rl.so;; esac
  MI_DMRECON_LIB=$PWD/$L timeout -s KILL 300 python bench.py $AB > $O/bench_$V.json 2> $O/bench_$V.err
  line $O/bench_$V.json
done
