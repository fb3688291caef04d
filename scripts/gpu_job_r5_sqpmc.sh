#!/bin/bash
# round 5: SQ-level counters of the Winograd conv kernel (where the non-MFMA cycles of a chunk go): one rocprofv3 --pmc pass per group
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sqpmc; mkdir -p $O
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $O/sq_counters.txt
i=0
for pmc in "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $O/g$i -o t -- python3 $R/scripts/wino_only.py 2 conv1_2 conv3_2 > $O/g$i.log 2>&1
  f=$(find $O/g$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY' >> $O/summary.txt
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name']
    if 'wino_conv' not in k: continue
    acc[k[:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    for c,vals in v.items():
        # one row per dispatch per (xcc/se) dimension: sum per dispatch = total/len(dispatches)
        print(k, c, sum(vals)/2.0)
PY
done
cat $O/summary.txt
tail -3 $O/g1.log
