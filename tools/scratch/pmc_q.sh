#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -oE "SQ_[A-Z_0-9]+" | sort -u | tr '\n' ' ' | head -c 6000 > $R/gpurun_out/sq_counters.txt
cd $R
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_FLAT SQ_INSTS_FLAT_LDS_ONLY SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT"; do
  tag=$(echo $set | md5sum | cut -c1-6)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmcq_$tag -o bench -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-two-in-flight --no-host-pipeline > /dev/null 2>&1
  python3 - <<PY
import csv,collections
tot=collections.defaultdict(lambda: collections.Counter())
try:
    for row in csv.DictReader(open('/tmp/pmcq_$tag/bench_counter_collection.csv')):
        n=row['Kernel_Name']
        for k in ('jit_quotient','jit_perm_rows','k_row_sponges','k_reduce_openings_quad'):
            if k in n: tot[k][row['Counter_Name']]+=float(row['Counter_Value'])
    for k,v in tot.items(): print(k, {a:'%.3g'%b for a,b in v.items()})
except Exception as e: print('ERR',e)
PY
done
