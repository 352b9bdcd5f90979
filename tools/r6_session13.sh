#!/bin/bash
# Runs ON THE GPU BOX: digests of the C3 maps under several builds; where two builds differ, by how much.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=gpurun_out/r6m
mkdir -p $O
for L in "-" $LIBS; do
  T=$( [ "$L" = "-" ] && echo new || basename $L .so | sed 's/libmi_dmrecon_//' )
  MI_DMRECON_LIB=$( [ "$L" = "-" ] && echo "" || echo $R/$L ) timeout -s KILL 300 python tools/maps_digest.py C3 5 > $O/digest_$T.json 2> $O/digest_$T.err; cut -c1-420 $O/digest_$T.json; tail -2 $O/digest_$T.err
  MI_DMRECON_LIB=$( [ "$L" = "-" ] && echo "" || echo $R/$L ) timeout -s KILL 300 python tools/maps_diff.py dump /tmp/maps_$T.npz 2> $O/dump_$T.err
  [ "$T" != new ] && python tools/maps_diff.py cmp /tmp/maps_new.npz /tmp/maps_$T.npz | tee $O/diff_$T.json | cut -c1-1500
done
NOX="--no-cpu-baseline --no-one-call --distinct-scenes 0 --no-seed-variant"
for i in 1 2; do
  MI_BENCH_REGION_LOG=1 timeout -s KILL 200 python bench.py --steps 20 --warmup 5 --repeats 3 $NOX > $O/bench_new_$i.json 2> $O/bench_new_$i.err
  echo "new driver plan: $(python -c "import json,sys; d=json.loads(open('$O/bench_new_$i.json').read().strip().splitlines()[-1]); print(round(d['value'],1), [round(x) for x in d['repeats']])")"
  grep region $O/bench_new_$i.err | tail -1
done
timeout -s KILL 200 python bench.py --streams 1 --steps-per-call 1 --steps 20 --warmup 3 --repeats 3 $NOX > $O/lone_new.json 2> $O/lone_new.err
echo "new lone calls: $(python -c "import json,sys; d=json.loads(open('$O/lone_new.json').read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],2))")"
timeout -s KILL 600 python -m pytest tests/test_gpu_parity.py -x -q > $O/pytest_parity.log 2>&1; tail -3 $O/pytest_parity.log
