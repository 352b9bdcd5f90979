#!/bin/bash
# Runs ON THE GPU BOX: round 6, k_generate with the stamps through an LDS tile against the stamps straight from memory
# (-DMI_GEN_GLOBAL_STAMPS): digest, parity suite, driver's plan and lone calls, twice each.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=gpurun_out/r6n
mkdir -p $O
timeout -s KILL 300 python tools/maps_digest.py C3 5 > $O/digest_new.json 2> $O/digest_new.err; cut -c1-200 $O/digest_new.json; tail -2 $O/digest_new.err
timeout -s KILL 600 python -m pytest tests/test_gpu_parity.py -x -q > $O/pytest_parity.log 2>&1; tail -2 $O/pytest_parity.log
NOX="--no-cpu-baseline --no-one-call --distinct-scenes 0 --no-seed-variant"
for L in "-" "build/libmi_dmrecon_genold.so" "-" "build/libmi_dmrecon_genold.so"; do
  T=$( [ "$L" = "-" ] && echo new || basename $L .so | sed 's/libmi_dmrecon_//' )_$RANDOM
  MI_DMRECON_LIB=$( [ "$L" = "-" ] && echo "" || echo $R/$L ) MI_BENCH_REGION_LOG=1 timeout -s KILL 200 python bench.py --steps 20 --warmup 5 --repeats 3 $NOX > $O/bench_$T.json 2> $O/bench_$T.err
  echo "$T driver plan: $(python -c "import json,sys; d=json.loads(open('$O/bench_$T.json').read().strip().splitlines()[-1]); print(round(d['value'],1), [round(x) for x in d['repeats']])")"
  grep region $O/bench_$T.err | tail -1
  MI_DMRECON_LIB=$( [ "$L" = "-" ] && echo "" || echo $R/$L ) timeout -s KILL 200 python bench.py --streams 1 --steps-per-call 1 --steps 20 --warmup 3 --repeats 3 $NOX > $O/lone_$T.json 2> $O/lone_$T.err
  echo "$T lone calls: $(python -c "import json,sys; d=json.loads(open('$O/lone_$T.json').read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],2))")"
done
