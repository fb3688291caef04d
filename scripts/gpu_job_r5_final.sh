# round 5, evidence on the final code: bench lines (C2 / C4 / C5 / conv_algo=direct / force-dist), kernel trace of the bench, per-layer
# times of the Winograd and the direct fp32 convs, Winograd phase clocks, PMC passes over the Winograd kernels (one counter group per
# pass; never combined with other trace domains), then the whole GPU suite with its printed agreement rates and the smoke run
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5final
mkdir -p $O
cd $R
(timeout -s ABRT 400 python -X faulthandler bench.py --steps 20 --warmup 5 > $O/bench_c2.json 2> $O/bench_c2.err; echo rc=$? >> $O/bench_c2.err)
(timeout -s ABRT 300 python -X faulthandler bench.py --steps 20 --warmup 5 --conv-algo direct --no-cpu-baseline --no-other-precisions > $O/bench_c2_direct.json 2> $O/bench_c2_direct.err)
for w in c4 c5; do
  (timeout -s ABRT 300 python -X faulthandler bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err; echo rc=$? >> $O/bench_$w.err)
done
(timeout -s ABRT 300 python -X faulthandler bench.py --gpus 1 --force-dist --steps 20 --warmup 3 --no-cpu-baseline --no-other-precisions > $O/bench_c2_force_dist.json 2> $O/bench_fd.err)
timeout 300 python scripts/layer_times_wino.py 2>&1 | grep -v amdgpu.ids > $O/layer_times_f32.txt
NAFAE_LIB=$R/nafae_amd/csrc/libnafae_hip_exp.so timeout 300 python scripts/wino_stamps.py 2>&1 | grep -v amdgpu.ids > $O/wino_stamps.txt
cd /tmp; export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o t -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-precisions > $O/bench_under_rocprof.json 2> $O/prof_bench.err
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_wino -o t -- python3 $R/scripts/wino_only.py 5 > $O/prof_wino.log 2>&1
for pmc in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $pmc | tr ' ' '_')
  timeout 200 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $O/pmc_wino_$n -o t -- python3 $R/scripts/wino_only.py 3 > $O/pmc_wino_$n.log 2>&1
done
cd $R
(timeout 2400 python -m pytest tests -q -m gpu -s --maxfail=10 > $O/gpu_all_verbose.log 2>&1; echo rc=$? >> $O/gpu_all_verbose.log)
grep -E "passed|failed|^FAILED|^ERROR|rc=" $O/gpu_all_verbose.log | tail -6
grep -E "accuracy|^\[[Cc][245]|^\[wino|bit-identical|differ in the last bits" $O/gpu_all_verbose.log > $O/gpu_tests_summary.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python - <<PY
import json
for n in ("c2","c2_direct","c4","c5","c2_force_dist"):
    try:
        d=json.loads(open("$O/bench_%s.json"%n).read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], d.get("roofline",{}).get("frac"), d.get("detector",{}).get("mfma_frac"), {k:v["value"] for k,v in d.get("modes",{}).items()})
    except Exception as e: print(n, "ERR", e)
PY
