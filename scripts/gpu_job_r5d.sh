set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5d
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_stress.py tests/test_gpu_configs.py -m gpu -q -x 2>&1 | tail -3 | tee $O/tests.log
for prec in bf16 bf16x3; do timeout 200 python scripts/layer_times.py $prec 2>&1 | grep -v amdgpu.ids | tee -a $O/layers.txt; done
timeout 600 python bench.py --precision bf16 --steps 20 --warmup 3 --no-cpu-baseline --no-other-precisions 2>$O/bench_bf16.err | tail -1 > $O/bench_bf16.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5d/bench_bf16.json"))
print("bf16", d.get("value"), d.get("ms_per_step"), d.get("stage_ms"))
PY
