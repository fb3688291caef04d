# round 4, final evidence: bench lines, kernel traces, PMC passes (scripts/gpu_job_r4_prof.sh), per-layer times of the three arithmetic
# modes, phase stamps of the similarity kernels, JPEG timing, then the whole GPU suite with its printed agreement rates
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4final
mkdir -p $O
cd $R
bash scripts/gpu_job_r4_prof.sh 2>&1 | tail -12
cd $R
timeout 300 python scripts/layer_times_f32.py 2>&1 | grep -v amdgpu.ids > $O/layer_times_f32.txt
timeout 300 python scripts/layer_times.py bf16x3 2>&1 | grep -v amdgpu.ids > $O/layer_times_bf16x3.txt
timeout 300 python scripts/layer_times.py bf16 2>&1 | grep -v amdgpu.ids > $O/layer_times_bf16.txt
for r in 0 14; do timeout 200 python scripts/jpeg_time.py $r 2>&1 | grep -v amdgpu.ids >> $O/jpeg_time.txt; done
export NAFAE_LIB=$R/nafae_amd/csrc/libnafae_hip_exp.so
timeout 200 python scripts/simplanes_time.py c5 bf16x3 f16 2>&1 | grep -v amdgpu.ids > $O/simplanes_stamps.txt
for c in c2 c4 c5; do
  timeout 120 python scripts/simplanes_hist.py $c none f16 2>&1 | grep -v amdgpu.ids >> $O/sim_narrow_hist.txt
done
for c in c2 c5; do SIM_LENS=hist NAFAE_SIM_NARROW=1 timeout 120 python scripts/simplanes_time.py $c f16 2>&1 | grep -v amdgpu.ids >> $O/sim_narrow_hist.txt; done
timeout 120 python scripts/simfused_stamps.py c5 hist 2>&1 | grep -v amdgpu.ids > $O/sim_live_stamps_c5.txt
timeout 120 python scripts/simfused_stamps.py c2 hist 2>&1 | grep -v amdgpu.ids > $O/sim_live_stamps_c2.txt
unset NAFAE_LIB
(timeout 2400 python -m pytest tests -q -m gpu -s --maxfail=10 > $O/gpu_all_verbose.log 2>&1; echo rc=$? >> $O/gpu_all_verbose.log)
grep -E "passed|failed|^FAILED|^ERROR|rc=" $O/gpu_all_verbose.log | tail -6
grep -E "accuracy|^\[[Cc][245]|bit-identical|differ in the last bits" $O/gpu_all_verbose.log > $O/gpu_tests_summary.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
