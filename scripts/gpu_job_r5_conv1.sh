#!/bin/bash
# round 5: what bounds conv1_1 (standalone launches, rocprofv3 kernel trace + SQ counters)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/conv1; mkdir -p $O
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o t -- python3 $R/scripts/conv1_only.py 5 > $O/kt.log 2>&1
python3 - <<PY
import csv
for r in csv.DictReader(open("$O/kt/t_kernel_stats.csv")):
    if 'conv1' in r['Name']: print(r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, 'us')
PY
i=0
for pmc in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $O/g$i -o t -- python3 $R/scripts/conv1_only.py 2 > $O/g$i.log 2>&1
  f=$(find $O/g$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name']
    if 'conv1' not in k: continue
    acc[k[:48]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    for c,vals in v.items(): print(k, c, sum(vals)/len(vals), len(vals))
PY
done
