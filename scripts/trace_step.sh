#!/bin/bash
# kernel timeline of the graph-replayed step: rocprofv3 --kernel-trace -> gpurun_out/trace/kernel_trace.csv (start/end per dispatch)
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trace_out
rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_out -- python3 $root/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-h2d "$@" > /tmp/trace.log 2>&1
tail -1 /tmp/trace.log
mkdir -p $root/gpurun_out/trace
f=$(ls /tmp/trace_out/*/*kernel_trace.csv | head -1)
python3 - "$f" "$root/gpurun_out/trace/kernel_trace_small.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[-12000:]          # the last ~4 steps
t0 = int(rows[0]['Start_Timestamp'])
with open(sys.argv[2], 'w') as f:
    f.write('start_ns,end_ns,queue,stream,name\n')
    for r in rows:
        f.write(f"{int(r['Start_Timestamp'])-t0},{int(r['End_Timestamp'])-t0},{r.get('Queue_Id','')},{r.get('Stream_Id','')},{r['Kernel_Name'][:90].replace(',',';')}\n")
PY
ls -la $root/gpurun_out/trace/
