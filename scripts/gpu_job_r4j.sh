set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4j
mkdir -p $O
cd $R
export NAFAE_LIB=$R/nafae_amd/csrc/libnafae_hip_exp.so
for c in c2 c4 c5; do
  SIM_LENS=hist timeout 120 python scripts/simplanes_time.py $c f16 2>&1 | grep -v amdgpu.ids | tee -a $O/narrow_stamps.txt
done
timeout 120 python scripts/simfused_stamps.py c2 hist 2>&1 | grep -v amdgpu.ids | tee -a $O/live_stamps.txt
timeout 120 python scripts/simfused_stamps.py c5 hist 2>&1 | grep -v amdgpu.ids | tee -a $O/live_stamps.txt
