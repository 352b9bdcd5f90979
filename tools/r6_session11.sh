#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): round 6, eleventh session -- the column-pair footprint elements as the product's layout:
# the whole GPU suite on it, then the driver's plan and lone calls against the 16-byte records of rounds 2-5 (-DMI_QUAD_RECORDS).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=gpurun_out/r6k
mkdir -p $O
( timeout -s KILL 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "rc $?" >> $O/pytest_gpu.log ); tail -6 $O/pytest_gpu.log
NOX="--no-cpu-baseline --no-one-call --distinct-scenes 0 --no-seed-variant"
for L in "" "build/libmi_dmrecon_quad16.so" "" "build/libmi_dmrecon_quad16.so"; do
  T=$( [ -z "$L" ] && echo new || basename $L .so | sed 's/libmi_dmrecon_//' )_$RANDOM
  MI_DMRECON_LIB=$( [ -z "$L" ] && echo "" || echo $R/$L ) MI_BENCH_REGION_LOG=1 timeout -s KILL 200 python bench.py --steps 20 --warmup 5 --repeats 3 $NOX > $O/bench_$T.json 2> $O/bench_$T.err
  echo "$T driver plan: $(python -c "import json,sys; d=json.loads(open('$O/bench_$T.json').read().strip().splitlines()[-1]); print(round(d['value'],1), [round(x) for x in d['repeats']])")"
  grep region $O/bench_$T.err | tail -1
  MI_DMRECON_LIB=$( [ -z "$L" ] && echo "" || echo $R/$L ) timeout -s KILL 200 python bench.py --streams 1 --steps-per-call 1 --steps 20 --warmup 3 --repeats 3 $NOX > $O/lone_$T.json 2> $O/lone_$T.err
  echo "$T lone calls: $(python -c "import json,sys; d=json.loads(open('$O/lone_$T.json').read().strip().splitlines()[-1]); r=d['roofline']['per_kernel']; print(round(d['value'],1), round(d['ms_per_step'],2), 'bulk ms/step', round(r['k_optimize<1> (host-visible rounds)']['avg_launch_ms']*r['k_optimize<1> (host-visible rounds)']['launches']/60,2), 'front', round(r['k_tail + k_front (tail rounds)']['k_front_ms']/60,2))")"
done
du -sh $O
