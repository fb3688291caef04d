#!/bin/bash
mkdir -p gpurun_out/r2m
O=gpurun_out/r2m
(timeout 900 python -m pytest tests/test_gpu_bf16.py -q -m gpu -x > $O/t.log 2>&1; echo rc=$? >> $O/t.log); tail -3 $O/t.log
python scripts/layer_times.py 2>&1 | grep -v amdgpu.ids | tee $O/layers.log
R=$PWD; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o t -- python3 $R/scripts/layer_times.py > /dev/null 2>&1 < /dev/null
cd $R; python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/r2m/prof/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "fixup" in r["Name"] or "run_sk" in r["Name"]: print(r["Name"][:80], r["Calls"], round(float(r["AverageNs"])/1e3,1))
PY
