#!/bin/bash
mkdir -p gpurun_out/r2j
O=gpurun_out/r2j
(timeout 300 python scripts/layer_times_f32.py > $O/layers_f32.log 2>&1; echo rc=$? >> $O/layers_f32.log); grep -v amdgpu.ids $O/layers_f32.log
(timeout 300 python scripts/layer_times.py > $O/layers.log 2>&1; echo rc=$? >> $O/layers.log); grep -v amdgpu.ids $O/layers.log
R=$PWD; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_bf16x3 -o t -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --precision bf16x3 --no-other-precisions > $R/$O/bench_bf16x3_prof.json 2> $R/$O/bench_bf16x3_prof.err < /dev/null
cd $R
python3 - <<'PY'
import csv,glob,json
f=glob.glob("gpurun_out/r2j/prof_bf16x3/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:22]: print(r["Name"][:70].replace("(anonymous namespace)::",""), r["Calls"], round(float(r["AverageNs"])/1e3,1), r["Percentage"])
b=json.load(open("gpurun_out/r2j/bench_bf16x3_prof.json")); print(b["value"], b["ms_per_step"], b["stage_ms"])
PY
