set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4g
mkdir -p $O
cd $R
(timeout 600 python -m pytest tests/test_gpu_jpeg.py -q -m gpu > $O/gpu_jpeg.log 2>&1; echo rc=$? >> $O/gpu_jpeg.log); tail -2 $O/gpu_jpeg.log
timeout 300 python scripts/jpeg_time.py 0 2>&1 | grep -v amdgpu.ids | tee $O/jpeg_time.txt
timeout 300 python scripts/jpeg_time.py 14 2>&1 | grep -v amdgpu.ids | tee -a $O/jpeg_time.txt
export NAFAE_LIB=$R/nafae_amd/csrc/libnafae_hip_exp.so
for k in 0 1 0 1; do
  echo "== NAFAE_F32_KORDER=$k" | tee -a $O/korder.txt
  NAFAE_F32_KORDER=$k timeout 300 python scripts/layer_times_f32.py 2>&1 | grep -v amdgpu.ids | tee -a $O/korder.txt
done
