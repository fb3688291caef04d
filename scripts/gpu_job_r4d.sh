set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4d
mkdir -p $O
cd $R
(timeout 900 python -m pytest tests/test_gpu_simplanes.py tests/test_gpu_simmax.py -q -m gpu --maxfail=20 > $O/gpu_sim.log 2>&1; echo rc=$? >> $O/gpu_sim.log)
grep -E "passed|failed|^FAILED|^ERROR|rc=" $O/gpu_sim.log | tail -25
for w in c5 c4 c2; do
  timeout 300 python scripts/simplanes_time.py $w none bf16x3 f16 2>&1 | grep -v amdgpu.ids | tee -a $O/simplanes_time.txt
done
export NAFAE_LIB=$R/nafae_amd/csrc/libnafae_hip_exp.so
timeout 300 python scripts/simplanes_time.py c5 bf16x3 f16 2>&1 | grep -v amdgpu.ids | tee $O/simplanes_stamps.txt
for w in c5 c2; do
  timeout 200 python scripts/simplanes_hist.py $w none f16 2>&1 | grep -v amdgpu.ids | tee -a $O/hist_routes.txt
  NAFAE_SIM_LIVE_MAX=0 timeout 200 python scripts/simplanes_hist.py $w none f16 bf16x3 2>&1 | grep -v amdgpu.ids | tee -a $O/hist_routes.txt
done
