#!/bin/bash
# GPU box: rocprofv3 kernel trace (every launch with its duration) of one bench run -> gpurun_out/trace_<tag>/
tag=${1:-x}; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/trace_$tag
mkdir -p $out
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trace_$tag -o bench -- python bench.py --no-cpu-baseline --no-two-in-flight --no-host-pipeline "$@" > $out/bench.log 2>&1
cp /tmp/trace_$tag/bench_kernel_stats.csv $out/ 2>/dev/null
python3 - <<PY
import csv
rows=list(csv.DictReader(open('/tmp/trace_$tag/bench_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
with open('$out/launches.csv','w') as f:
    t0=int(rows[0]['Start_Timestamp'])
    for r in rows:
        f.write('%d,%d,%s,%s,%s\n'%(int(r['Start_Timestamp'])-t0,int(r['End_Timestamp'])-int(r['Start_Timestamp']),r['Kernel_Name'][:60].replace(',',';'),r.get('Grid_Size_X', r.get('Grid_Size','')),r.get('Workgroup_Size_X', r.get('Workgroup_Size',''))))
PY
gzip -f $out/launches.csv
ls -la $out
