set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5c
mkdir -p $O
cd $R
for rep in 1 2; do
for v in v0 v1 v4 v8 v15; do
  for prec in bf16 bf16x3; do
    NAFAE_LIB=$R/nafae_amd/csrc/variants/libnafae_hip_$v.so PREC=$prec timeout 200 python3 scripts/conv_times.py c12 c21 c22 2>&1 | grep -v amdgpu.ids | tee -a $O/variants_patch.txt
  done
done
done
