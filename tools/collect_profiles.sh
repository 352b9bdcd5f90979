#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the round's committed evidence for the C3 bench -- bench lines, rocprofv3 kernel stats and
# PMC passes AT THE CALL PLANS THE LINES ARE MEASURED AT: the driver's command (--steps 20 --warmup 5: 4 host threads x 5 steps
# per call, merged into one 400-view batch) and one lone 20-view call.  Everything lands in gpurun_out/$TAG/;
# tools/summarize_profiles.py turns it into profiles/<tag>_*.
# --pmc passes are separate runs with --kernel-trace only (never with sys/runtime/hip traces), each under its own timeout: a
# counter group the hardware cannot collect makes rocprofv3 abort and then hang in its finalisation.
TAG=${1:-r6}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
MI_BENCH_REGION_LOG=1 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err      # the driver's command
python bench.py --streams 1 --steps-per-call 1 --steps 20 --repeats 3 --no-cpu-baseline --no-one-call --distinct-scenes 0 --no-seed-variant > $OUT/bench_1thread.json 2> $OUT/bench_1thread.err
NOX="--no-cpu-baseline --no-one-call --distinct-scenes 0 --no-seed-variant"      # (the plain line only: nothing but the default form in a profiled run)
B1="python $R/bench.py --steps 6 --warmup 1 --repeats 1 --streams 1 --steps-per-call 1 $NOX"
B3="python $R/bench.py --steps 20 --warmup 5 --repeats 2 $NOX"
cd /tmp
# kernel traces (no --stats: its summary would count the warm-up calls, which run as differently shaped batches) -- cut to the TIMED
# REGIONS by the k_region_mark dispatches bench.py brackets them with (tools/trace_regions.py), the bench line of the same run next
# to them, and the roofline fractions recomputed from both (tools/roofline_check.py)
timeout -s KILL 150 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/s1 -o bench -- $B1 > $R/$OUT/s1.log 2>&1
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/s3 -o bench -- $B3 > $R/$OUT/s3.log 2>&1
cd $R
for S in s1 s3; do
  F=$(find $OUT/$S -name "*kernel_trace.csv" | head -1)
  [ "$S" = s1 ] && ST=6 || ST=20
  if [ -n "$F" ]; then
    python tools/trace_regions.py $F $ST --csv=$OUT/${S}_timed_regions_stats.csv > $OUT/${S}_timed_regions.json 2> $OUT/${S}_timed_regions.err
    grep "^{" $OUT/$S.log | tail -1 > $OUT/${S}_line.json
    python tools/roofline_check.py $OUT/${S}_timed_regions_stats.csv $OUT/${S}_line.json > $OUT/${S}_roofline_check.md 2> $OUT/${S}_roofline_check.err
  fi
done
tail -8 $OUT/s3_roofline_check.md
cd /tmp
# PMC at the plan of the lone call (one step per call, one host thread: 3 calls) ...
BP="python $R/bench.py --steps 2 --warmup 1 --repeats 1 --streams 1 --steps-per-call 1 $NOX"
# ... and at the DRIVER's plan: 4 host threads x 5 steps per call = one 400-view batch per region (warm-up: one more)
BQ="python $R/bench.py --steps 20 --warmup 1 --repeats 1 $NOX"
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  D=$R/$OUT/pmc_$(echo $C | tr ' ' '+')
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D -o bench -- $BP > $D.log 2>&1
done
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  D=$R/$OUT/pmcd_$(echo $C | tr ' ' '+')
  timeout -s KILL 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D -o bench -- $BQ > $D.log 2>&1
done
cd $R
# keep the merge small: drop the per-dispatch traces of the stats runs (the stats CSV is the summary)
find $OUT/s1 $OUT/s3 -name "*kernel_trace.csv" -delete
find $OUT -name "*_kernel_trace.csv" -path "*pmc*" -delete
du -sh $OUT
cat $OUT/bench_driver.json $OUT/bench_1thread.json | cut -c1-300
# a round trace of one lone C3 call, lone calls of a rank's share, the other configurations' lines
timeout -s KILL 120 python tools/trace_c3.py > $OUT/round_trace_c3.txt 2>&1
[ -n "$SKIP_EXTRAS" ] && exit 0                     # (a short collection: the C3 lines, kernel stats and counters only)
timeout -s KILL 300 python tools/lone_calls.py C3 12 > $OUT/lone_calls.json 2> $OUT/lone_calls.err
timeout -s KILL 300 python bench.py --config C2 --steps 20 --warmup 3 > $OUT/bench_c2.json 2> $OUT/bench_c2.err
timeout -s KILL 600 python bench.py --config C5 --steps 4 --warmup 1 --repeats 3 --streams 2 --steps-per-call 1 > $OUT/bench_c5.json 2> $OUT/bench_c5.err
cut -c1-300 $OUT/lone_calls.json; tail -c 600 $OUT/bench_c5.json
