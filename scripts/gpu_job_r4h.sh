set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4h
mkdir -p $O
cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 scripts/micro/buffer_lds_oob.hip -o /tmp/buffer_lds_oob 2>/dev/null && timeout 60 /tmp/buffer_lds_oob | tee $O/micro_oob.txt
timeout 600 python scripts/conv4_check.py 2>&1 | grep -v amdgpu.ids | tee $O/conv4_check.txt
export NAFAE_LIB=$R/nafae_amd/csrc/libnafae_hip_exp.so
for k in 1 0 1 0; do
  echo "== NAFAE_F32_CONV4=$k" | tee -a $O/conv4_layers.txt
  NAFAE_F32_CONV4=$k timeout 300 python scripts/layer_times_f32.py 2>&1 | grep -v amdgpu.ids | tee -a $O/conv4_layers.txt
done
