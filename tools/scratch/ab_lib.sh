#!/bin/bash
# same-box A/B of two builds of the library: ab_lib.sh <other .so under lurk_amd/>
R=$GRAFT_REPO_ROOT; cd $R
run() { env $1 python bench.py --lanes 1 --steps 10 --no-cpu-baseline --no-host-pipeline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['config']['stages_ms']
print('$1', round(d['ms_per_step'],3), 'lde', round(s['lde'],3), 'leaves', round(s['merkle_leaves'],3))"; }
for i in 1 2; do run X=default; run LURKHIP_LIB_PATH=$R/lurk_amd/$1; done
