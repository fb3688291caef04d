set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3u
mkdir -p $O
cd $R
(timeout 2400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_simmax.py tests/test_gpu_stress.py tests/test_gpu_configs.py tests/test_gpu_exact_dp.py tests/test_gpu_model.py -q -m gpu --maxfail=20 > $O/tests.log 2>&1; echo rc=$? >> $O/tests.log)
grep -E "passed|failed|^FAILED|^ERROR|rc=" $O/tests.log | tail
python scripts/sim_sweep_live.py c5 2>&1 | grep -v amdgpu | head -5
