set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4u
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_stress.py tests/test_gpu_widened.py tests/test_gpu_configs.py -m gpu -q -x 2>&1 | tail -6 | tee $O/tests.log
export NAFAE_LIB=$R/nafae_amd/csrc/libnafae_hip_exp.so
for k in kernel inline kernel inline; do
  echo "== NAFAE_SK_FIXUP=$k f32" | tee -a $O/layers_f32.txt
  NAFAE_SK_FIXUP=$k timeout 300 python scripts/layer_times_f32.py 2>&1 | grep -v amdgpu.ids | tee -a $O/layers_f32.txt
done
unset NAFAE_LIB
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>$O/bench_f32.err | tail -1 > $O/bench_f32.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4u/bench_f32.json"))
print("f32", d.get("value"), d.get("ms_per_step"), {k:v["value"] for k,v in d.get("modes",{}).items()})
PY
