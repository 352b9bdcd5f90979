#!/bin/bash
# Runs ON THE GPU BOX (through gpurun).  Session r4j.
export TMPDIR=/tmp
OUT=gpurun_out/r4j
mkdir -p $OUT
timeout -s KILL 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --one-call-n 30 2>$OUT/bench.err > $OUT/bench.json
python - $OUT/bench.json <<'PY'
import sys, json
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
oc = d["one_call"]
print("value %.1f" % d["value"], "repeats", [round(x) for x in d["repeats"]], "one_call %.2f ms bulk %.2f front %.2f plan %.2f" % (oc["ms_per_call"], oc["ms_bulk_kernel"], oc["ms_front_kernel"], oc["ms_host_planning"]))
PY
cd /tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/s1 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --repeats 1 --streams 1 --steps-per-call 1 --no-cpu-baseline --no-one-call > $GRAFT_REPO_ROOT/$OUT/s1.log 2>&1
cd $GRAFT_REPO_ROOT
rm -f $OUT/s1/*kernel_trace.csv
cut -c1-160 $OUT/s1/bench_kernel_stats.csv | head -8
