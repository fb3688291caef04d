set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5g
mkdir -p $O
cd $R
(timeout 2400 python -m pytest tests -q -m gpu --maxfail=10 > $O/gpu_all.log 2>&1; echo rc=$? >> $O/gpu_all.log)
grep -E "passed|failed|^FAILED|^ERROR|rc=" $O/gpu_all.log | tail -8
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>$O/bench.err | tail -1 > $O/bench_f32_$i.json
done
python - <<'PY'
import json
for i in (1,2):
    d=json.load(open("gpurun_out/r5g/bench_f32_%d.json"%i))
    print("f32", d.get("value"), d.get("ms_per_step"), {k:v["value"] for k,v in d.get("modes",{}).items()})
PY
