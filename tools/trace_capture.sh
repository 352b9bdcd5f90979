#!/bin/bash
# GPU box: kernel trace of the bench for offline analysis -> gpurun_out/tr_<label>.csv
# usage: tools/trace_capture.sh LABEL STREAMS [ENV=VAL ...]
export TMPDIR=/tmp
mkdir -p gpurun_out
label=$1; streams=$2; shift 2
rm -rf /tmp/tr; env "$@" X=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python bench.py --steps 6 --warmup 1 --streams $streams --steps-per-call 1 --no-cpu-baseline > /tmp/tr.log 2>&1
python - "$label" <<'PY'
import csv, sys
rows = list(csv.DictReader(open('/tmp/tr/t_kernel_trace.csv')))
w = csv.writer(open('gpurun_out/tr_%s.csv' % sys.argv[1], 'w'))
w.writerow(['queue', 'name', 'start', 'end', 'grid'])
for r in rows:
    w.writerow([r['Queue_Id'], r['Kernel_Name'].split('(')[0].replace('void ', ''), r['Start_Timestamp'], r['End_Timestamp'], r['Grid_Size_X']])
PY
ls -la gpurun_out/tr_$label.csv
