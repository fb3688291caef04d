set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4v
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_ops.py tests/test_gpu_stress.py tests/test_gpu_configs.py tests/test_gpu_model.py -m gpu -q -x 2>&1 | tail -6 | tee $O/tests.log
export NAFAE_LIB=$R/nafae_amd/csrc/libnafae_hip_exp.so
for k in kernel inline kernel inline; do
  for prec in bf16 bf16x3; do
    echo "== NAFAE_SK_FIXUP=$k $prec" | tee -a $O/layers.txt
    NAFAE_SK_FIXUP=$k timeout 300 python scripts/layer_times.py $prec 2>&1 | grep -v amdgpu.ids | grep -v "^conv1\|^conv2" | tee -a $O/layers.txt
  done
done
unset NAFAE_LIB
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>$O/bench_f32.err | tail -1 > $O/bench_f32.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4v/bench_f32.json"))
print("f32", d.get("value"), d.get("ms_per_step"), {k:v["value"] for k,v in d.get("modes",{}).items()})
PY
