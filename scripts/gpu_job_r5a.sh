set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5a
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_configs.py tests/test_gpu_model.py tests/test_gpu_bf16.py -m gpu -q -x -s 2>&1 | grep -E "passed|failed|FAILED|c2 +bf16|C2 bf16|bf16 " | tail -12 | tee $O/tests.log
NAFAE_LIB=$R/nafae_amd/csrc/libnafae_hip_exp.so timeout 900 python tests/dispatch_worker.py $O/dispatch_table.json > $O/dispatch.log 2>&1; echo "dispatch rc=$?"
for i in 1 2; do
timeout 600 python bench.py --precision bf16 --steps 20 --warmup 3 --no-cpu-baseline --no-other-precisions 2>$O/bench_bf16.err | tail -1 > $O/bench_bf16_$i.json
done
python - <<'PY'
import json
for i in (1,2):
    d=json.load(open("gpurun_out/r5a/bench_bf16_%d.json"%i))
    print("bf16", d.get("value"), d.get("ms_per_step"), d.get("stage_ms"))
PY
