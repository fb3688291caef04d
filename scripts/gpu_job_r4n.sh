set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4n
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_bf16.py -m gpu -q -x -k "one_wave" 2>&1 | tail -5 | tee $O/tests_conv.log
export NAFAE_LIB=$R/nafae_amd/csrc/libnafae_hip_exp.so
for k in 0 1; do
  for prec in bf16 bf16x3; do
    NAFAE_CONV4=$k timeout 300 python scripts/conv_occupancy.py $prec 2>&1 | grep -v amdgpu.ids | tee -a $O/occupancy.txt
  done
done
unset NAFAE_LIB
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 | tee $O/tests_all.log
