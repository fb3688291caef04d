set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2
cd $R
(timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r2/gpu_all.log 2>&1; echo rc=$? >> gpurun_out/r2/gpu_all.log)
grep -E "^\[C|passed|failed|^FAILED|^ERROR|rc=" gpurun_out/r2/gpu_all.log | tail -30
(timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/r2/bench.json 2> gpurun_out/r2/bench.err; echo rc=$? >> gpurun_out/r2/bench.err)
tail -c 300 gpurun_out/r2/bench.err
cd /tmp; export TMPDIR=/tmp
# kernel trace of the benched command (f32 headline + siblings), no CPU baseline
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2/prof_bench -o t -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/r2/bench_under_rocprof.json 2> $R/gpurun_out/r2/prof_bench.err
# sim kernels alone at C5 (histogram lengths and dense) and C2: kernel trace + PMC passes
for c in "c5 hist" "c5 dense" "c2 hist"; do set -- $c
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2/sim_$1_$2 -o t -- python3 $R/scripts/sim_only.py $1 $2 20 > $R/gpurun_out/r2/sim_$1_$2.log 2>&1
  for pmc in FETCH_SIZE WRITE_SIZE; do
    timeout 120 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $R/gpurun_out/r2/pmc_$1_$2_$pmc -o t -- python3 $R/scripts/sim_only.py $1 $2 5 > $R/gpurun_out/r2/pmc_$1_$2_$pmc.log 2>&1
  done
done
ls $R/gpurun_out/r2 | head -40
