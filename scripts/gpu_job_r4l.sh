set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4l
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_bf16.py -m gpu -q -x -k "conv" 2>&1 | tail -15 | tee $O/tests_conv.log
export NAFAE_LIB=$R/nafae_amd/csrc/libnafae_hip_exp.so
for k in 1 0; do
  for prec in bf16 bf16x3; do
    echo "== NAFAE_CONV4=$k $prec" | tee -a $O/layers.txt
    NAFAE_CONV4=$k timeout 300 python scripts/layer_times.py $prec 2>&1 | grep -v amdgpu.ids | tee -a $O/layers.txt
  done
done
unset NAFAE_LIB
timeout 600 python -m pytest tests/test_gpu_simmax.py tests/test_gpu_simplanes.py tests/test_gpu_model.py -m gpu -q -x 2>&1 | tail -8 | tee $O/tests_sim.log
timeout 600 python bench.py --precision bf16 --steps 10 --warmup 3 2>$O/bench_bf16.err | tail -1 > $O/bench_bf16.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4l/bench_bf16.json"))
print("bf16", d.get("value"), d.get("ms_per_step"))
PY
