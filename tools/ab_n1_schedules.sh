cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu-baseline --no-host-pipeline --steps 20 --warmup 3 "${@:2}" 2>/dev/null | python3 -c "
import json,sys
l=[x for x in sys.stdin.read().splitlines() if x.startswith('{')]
b=json.loads(l[-1]); print('$1', round(b['ms_per_step'],2), b.get('proofs_in_flight'), b['config'].get('rank_pipeline') is not None)"; }
for rep in 1 2; do
run "lanes 2 (default)"
run "lanes 3" --lanes 3
run "lanes 4" --lanes 4
run "pipelined d3" --rank-pipeline --rank-pipeline-depth 3
run "pipelined d2" --rank-pipeline
done
