set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3v
mkdir -p $O
cd $R
(timeout 1500 python -m pytest tests/test_gpu_simmax.py tests/test_gpu_stress.py tests/test_gpu_ops.py -q -m gpu --maxfail=20 > $O/tests.log 2>&1; echo rc=$? >> $O/tests.log)
grep -E "passed|failed|^FAILED|^ERROR|rc=" $O/tests.log | tail
export NAFAE_LIB=$R/nafae_amd/csrc/libnafae_hip_exp.so
for m in 32 100000; do
  echo "== NAFAE_SIM_LIVE_MAX=$m"
  NAFAE_SIM_LIVE_MAX=$m python scripts/sim_sweep_live.py c5 2>&1 | grep -v amdgpu
  NAFAE_SIM_LIVE_MAX=$m python scripts/sim_sweep_live.py c2 2>&1 | grep -v amdgpu
  NAFAE_SIM_LIVE_MAX=$m python scripts/sim_sweep_live.py c4 2>&1 | grep -v amdgpu
done
