# round 4: bench lines + kernel traces + PMC passes of the similarity kernels (one counter group per pass; never combined with other
# trace domains).  The whole GPU suite runs in scripts/gpu_job_r4_tests.sh.
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4final
mkdir -p $O
cd $R
(timeout -s ABRT 300 python -X faulthandler bench.py --steps 20 --warmup 3 > $O/bench_c2.json 2> $O/bench_c2.err; echo rc=$? >> $O/bench_c2.err)
for w in c4 c5; do
  (timeout -s ABRT 240 python -X faulthandler bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err; echo rc=$? >> $O/bench_$w.err)
done
(timeout -s ABRT 240 python -X faulthandler bench.py --gpus 1 --force-dist --steps 20 --warmup 3 --no-cpu-baseline --no-other-precisions > $O/bench_c2_force_dist.json 2> $O/bench_fd.err)
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o t -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/prof_bench.err
for c in "c5 hist" "c5 dense" "c2 hist" "c4 hist"; do set -- $c
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/sim_$1_$2 -o t -- python3 $R/scripts/sim_only.py $1 $2 20 > $O/sim_$1_$2.log 2>&1
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/simloss_$1_$2 -o t -- python3 $R/scripts/simloss_only.py $1 $2 20 > $O/simloss_$1_$2.log 2>&1
  for pmc in FETCH_SIZE WRITE_SIZE; do
    timeout 120 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $O/pmc_$1_$2_$pmc -o t -- python3 $R/scripts/sim_only.py $1 $2 5 > $O/pmc_$1_$2_$pmc.log 2>&1
  done
done
cd $R
ls $O | wc -l
python - <<PY
import json
d=json.loads(open("$O/bench_c2.json").read().strip().splitlines()[-1])
print("C2", d["value"], d["ms_per_step"], d["roofline"]["frac"], {k:v["value"] for k,v in d.get("modes",{}).items()})
print("roofline_sim", d["roofline_sim"]["avg_ms"], d["roofline_sim"]["frac"])
for k,v in d["sim_loss_c5"].items():
    if isinstance(v, dict): print(k, "fwd_ms", v["fwd_ms"], "frac", v["fwd_hbm_frac"], "fwd_bwd_ms", v["fwd_bwd_ms"], v["fwd_bwd_hbm_frac"])
print("cpu", d.get("cpu_baseline",{}).get("value"))
PY
