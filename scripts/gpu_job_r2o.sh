#!/bin/bash
mkdir -p gpurun_out/r2o
O=$PWD/gpurun_out/r2o; R=$PWD
cd /tmp; export TMPDIR=/tmp
rocprofv3 --list-avail > $O/avail.txt 2>&1 < /dev/null
grep -c "" $O/avail.txt
for c in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-40)
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/p_$n -o t -- python3 $R/scripts/conv12_f32_only.py > $O/p_$n.log 2>&1 < /dev/null
done
cd $R
python3 - <<'PY'
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/r2o/p_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv3x3_kernel" not in r["Kernel_Name"]: continue
        k="128x64" if "128, 64" in r["Kernel_Name"] else "128x128"
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,c in agg.items():
    print(k, {a:round(sum(b)/len(b)) for a,b in sorted(c.items())})
PY
