# round 4, second job: the planes kernels + one-launch live-column kernel: tests, stand-alone timings, stamps
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4b
mkdir -p $O
cd $R
(timeout 1500 python -m pytest tests/test_gpu_simplanes.py tests/test_gpu_simmax.py tests/test_gpu_configs.py tests/test_gpu_exact_dp.py tests/test_gpu_model.py -q -m gpu --maxfail=40 --durations=5 > $O/gpu_sim.log 2>&1; echo rc=$? >> $O/gpu_sim.log)
grep -E "passed|failed|^FAILED|^ERROR|rc=" $O/gpu_sim.log | tail -45
grep -E "accuracy" $O/gpu_sim.log | head -8
for w in c5 c4 c2; do
  timeout 300 python scripts/simplanes_time.py $w none bf16x3 f16 2>&1 | tee -a $O/simplanes_time.txt
done
NAFAE_LIB=$R/nafae_amd/csrc/libnafae_hip_exp.so timeout 300 python scripts/simplanes_time.py c5 bf16x3 f16 2>&1 | tee $O/simplanes_stamps.txt
(timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-precisions > $O/bench_c2.json 2> $O/bench_c2.err; echo rc=$? >> $O/bench_c2.err)
python - <<PY
import json
d=json.loads(open("$O/bench_c2.json").read().strip().splitlines()[-1])
print("C2", d["value"], d["ms_per_step"])
print("roofline_sim", d["roofline_sim"]["avg_ms"], d["roofline_sim"]["frac"])
for k,v in d["sim_loss_c5"].items():
    if isinstance(v, dict): print(k, "fwd_ms", v["fwd_ms"], "frac", v["fwd_hbm_frac"], "fwd_bwd_ms", v["fwd_bwd_ms"], v["fwd_bwd_hbm_frac"])
print("sim_loss_only", d["sim_loss_only"]["fwd_ms"], d["sim_loss_only"]["fwd_bwd_ms"])
PY
