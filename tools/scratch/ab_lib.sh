#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
run() { env $1 $2 python bench.py --lanes 1 --steps 10 --no-cpu-baseline --no-host-pipeline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['config']['stages_ms']
print('$1 $2', round(d['ms_per_step'],3), 'lde', round(s['lde'],3), 'leaves', round(s['merkle_leaves'],3))"; }
run LURKHIP_NTT_FUSED=0 X=1
run LURKHIP_NTT_FUSED=0 LURKHIP_LIB_PATH=$R/lurk_amd/liblurkhip_slots5.so
run LURKHIP_NTT_FUSED=1 LURKHIP_LIB_PATH=$R/lurk_amd/liblurkhip_slots5.so
run LURKHIP_NTT_FUSED=0 X=1
run LURKHIP_NTT_FUSED=1 LURKHIP_LIB_PATH=$R/lurk_amd/liblurkhip_slots5.so
