set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4i
mkdir -p $O
cd $R
export NAFAE_LIB=$R/nafae_amd/csrc/libnafae_hip_exp.so
for c in c2 c4 c5; do
  timeout 120 python scripts/simplanes_hist.py $c none f16 2>&1 | grep -v amdgpu.ids | tee -a $O/narrow_hist.txt
done
unset NAFAE_LIB
timeout 1200 python -m pytest tests/test_gpu_simmax.py tests/test_gpu_simplanes.py tests/test_gpu_model.py tests/test_gpu_configs.py -m gpu -q -x 2>&1 | tail -15 | tee $O/tests.log
