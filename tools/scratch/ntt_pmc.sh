#!/bin/bash
# GPU box: PMC counters of the NTT passes of one 2^20 x 78 LDE (tools/lde_throughput.py)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAIT_INST_VMEM"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/nttpmc_$tag -o x -- python $R/tools/lde_throughput.py 20x78 > /dev/null 2>&1
  python3 - <<PY
import csv, collections
rows=list(csv.DictReader(open('/tmp/nttpmc_$tag/x_counter_collection.csv')))
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in rows:
    k=r['Kernel_Name'][:48]
    if 'ntt' not in k: continue
    agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
seen=set()
for r in rows:
    k=r['Kernel_Name'][:48]
    if 'ntt' in k:
        key=(k,r.get('Dispatch_Id'))
        if key not in seen: seen.add(key); cnt[k]+=1
for k,v in agg.items():
    print(k, 'launches', cnt[k], {c: '%.3g'%x for c,x in v.items()})
PY
done
