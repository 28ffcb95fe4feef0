cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_open_gpu.py tests/test_cpu_step_gpu.py -x -q -m gpu 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for slab in 1 0; do
LURKHIP_DOT_SLAB=$slab rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pd_$slab -o run -- python $GRAFT_REPO_ROOT/bench.py --lanes 1 --steps 4 --warmup 1 --no-cpu-baseline --no-host-pipeline > /tmp/pd_$slab.log 2>&1
python3 - <<PY
import csv, re
tot=0
for r in csv.DictReader(open('/tmp/pd_$slab/run_kernel_stats.csv')):
    if 'column_dot' in r['Name'] or 'dot_finish' in r['Name']:
        print('  slab=$slab', r['Name'][:60], r['Calls'], 'ms/proof=%.3f'%(float(r['TotalDurationNs'])/1e6/5))
PY
done
