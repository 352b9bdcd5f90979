#!/bin/bash
# GPU box: kernel traces of the multi-stream bench for offline overlap analysis -> gpurun_out/tr_<label>.csv
export TMPDIR=/tmp
mkdir -p gpurun_out
cap() { # label, env..., -- streams
  local label=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  rm -rf /tmp/tr; env "${envs[@]}" rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python bench.py --steps 9 --warmup 1 --streams $1 --no-cpu-baseline > /tmp/tr.log 2>&1
  python - "$label" <<'PY'
import csv, sys
rows = list(csv.DictReader(open('/tmp/tr/t_kernel_trace.csv')))
w = csv.writer(open('gpurun_out/tr_%s.csv' % sys.argv[1], 'w'))
w.writerow(['queue', 'name', 'start', 'end', 'grid'])
for r in rows:
    w.writerow([r['Queue_Id'], r['Kernel_Name'].split('(')[0].replace('void ', ''), r['Start_Timestamp'], r['End_Timestamp'], r['Grid_Size_X']])
PY
}
cap base3 X=1 -- 3
cap grid2048_3 MI_DMRECON_BULK_GRID=2048 -- 3
cap grid1024_3 MI_DMRECON_BULK_GRID=1024 -- 3
ls -la gpurun_out/tr_*.csv
