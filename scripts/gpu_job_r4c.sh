# round 4, third job: planes kernel with register staging + row masking; A/B vs LDS-DMA; few-live stamps
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4c
mkdir -p $O
cd $R
(timeout 900 python -m pytest tests/test_gpu_simplanes.py tests/test_gpu_simmax.py -q -m gpu --maxfail=20 > $O/gpu_sim.log 2>&1; echo rc=$? >> $O/gpu_sim.log)
grep -E "passed|failed|^FAILED|^ERROR|rc=" $O/gpu_sim.log | tail -25
for w in c5 c4 c2; do
  timeout 300 python scripts/simplanes_time.py $w none bf16x3 f16 2>&1 | grep -v amdgpu.ids | tee -a $O/simplanes_time.txt
done
export NAFAE_LIB=$R/nafae_amd/csrc/libnafae_hip_exp.so
timeout 300 python scripts/simplanes_time.py c5 bf16x3 f16 2>&1 | grep -v amdgpu.ids | tee $O/simplanes_stamps_regs.txt
NAFAE_SIM_STG=0 timeout 300 python scripts/simplanes_time.py c5 bf16x3 f16 2>&1 | grep -v amdgpu.ids | tee $O/simplanes_stamps_dma.txt
NAFAE_SIM_DBG=2 timeout 300 python scripts/simplanes_time.py c5 f16 2>&1 | grep -v amdgpu.ids | tee $O/simplanes_stamps_nomfma.txt
timeout 300 python scripts/simfused_stamps.py c5 hist 2>&1 | grep -v amdgpu.ids | tee $O/few_stamps_c5.txt
timeout 300 python scripts/simfused_stamps.py c2 hist 2>&1 | grep -v amdgpu.ids | tee $O/few_stamps_c2.txt
