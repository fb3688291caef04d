set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5b
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_widened.py tests/test_gpu_ops.py tests/test_gpu_jpeg.py -m gpu -q -x 2>&1 | tail -3 | tee $O/tests.log
for prec in bf16 bf16x3; do timeout 200 python scripts/layer_times.py $prec 2>&1 | grep "conv1_1" | tee -a $O/conv1.txt; done
timeout 200 python scripts/layer_times_f32.py 2>&1 | grep "conv1_1" | tee -a $O/conv1.txt
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>$O/bench.err | tail -1 > $O/bench_f32.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5b/bench_f32.json"))
print("f32", d.get("value"), d.get("ms_per_step"), {k:v["value"] for k,v in d.get("modes",{}).items()})
PY
