set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4x
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_widened.py -m gpu -q -x 2>&1 | tail -5 | tee $O/tests_first.log
export NAFAE_LIB=$R/nafae_amd/csrc/libnafae_hip_exp.so
for k in 0 1 0 1; do
  for prec in bf16 bf16x3; do
    echo "== NAFAE_CONV1_MFMA=$k NAFAE_CONV_SK_SMALL=$k $prec" | tee -a $O/layers.txt
    NAFAE_CONV1_MFMA=$k NAFAE_CONV_SK_SMALL=$k timeout 300 python scripts/layer_times.py $prec 2>&1 | grep -v amdgpu.ids | grep "conv1_1\|conv5\|rpn\|sum" | tee -a $O/layers.txt
  done
done
timeout 900 python tests/dispatch_worker.py $O/dispatch_table.json > $O/dispatch.log 2>&1; echo "dispatch rc=$?"; tail -2 $O/dispatch.log
unset NAFAE_LIB
(timeout 2400 python -m pytest tests -q -m gpu --maxfail=10 --durations=5 --deselect tests/test_gpu_dispatch.py > $O/gpu_all.log 2>&1; echo rc=$? >> $O/gpu_all.log)
grep -E "passed|failed|^FAILED|^ERROR|rc=" $O/gpu_all.log | tail -12
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>$O/bench_f32.err | tail -1 > $O/bench_f32.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4x/bench_f32.json"))
print("f32", d.get("value"), d.get("ms_per_step"), {k:v["value"] for k,v in d.get("modes",{}).items()})
PY
