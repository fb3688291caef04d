set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4e
mkdir -p $O
cd $R
NAFAE_LIB=$R/nafae_amd/csrc/libnafae_hip_exp.so timeout 900 python tests/dispatch_worker.py $O/dispatch_table.json > $O/dispatch.log 2>&1; echo "dispatch rc=$?"; tail -3 $O/dispatch.log
(timeout 2400 python -m pytest tests -q -m gpu --maxfail=30 --durations=8 --deselect tests/test_gpu_dispatch.py > $O/gpu_all.log 2>&1; echo rc=$? >> $O/gpu_all.log)
grep -E "passed|failed|^FAILED|^ERROR|rc=" $O/gpu_all.log | tail -12
for w in c5 c4 c2; do
  timeout 300 python scripts/simplanes_time.py $w none bf16x3 f16 2>&1 | grep -v amdgpu.ids | tee -a $O/simplanes_time.txt
done
(timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; echo rc=$? >> $O/bench_c2.err)
python - <<PY
import json
d=json.loads(open("$O/bench_c2.json").read().strip().splitlines()[-1])
print("C2", d["value"], d["ms_per_step"], {k:v["value"] for k,v in d.get("modes",{}).items()})
print("roofline_sim", d["roofline_sim"]["avg_ms"], d["roofline_sim"]["frac"])
for k,v in d["sim_loss_c5"].items():
    if isinstance(v, dict): print(k, "fwd_ms", v["fwd_ms"], "frac", v["fwd_hbm_frac"], "fwd_bwd_ms", v["fwd_bwd_ms"], v["fwd_bwd_hbm_frac"])
PY
