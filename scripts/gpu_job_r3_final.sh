# round 3: whole gpu suite, smoke, bench lines (C2 headline with cpu_baseline, C4, C5, streamed input, pipeline on/off in bf16),
# kernel traces and PMC passes (one counter group per pass; never combined with other trace domains)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3f
mkdir -p $O
cd $R
(timeout 2400 python -m pytest tests -q -m gpu --maxfail=30 --durations=10 > $O/gpu_all.log 2>&1; echo rc=$? >> $O/gpu_all.log)
grep -E "passed|failed|^FAILED|^ERROR|rc=" $O/gpu_all.log | tail -20
grep -E "^\[C[245]" $O/gpu_all.log > $O/gpu_tests_summary.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
(timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench_c2.json 2> $O/bench_c2.err; echo rc=$? >> $O/bench_c2.err)
for w in c4 c5; do
  (timeout 600 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err; echo rc=$? >> $O/bench_$w.err)
done
(timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --stream-input > $O/bench_c2_stream.json 2> $O/bench_c2_stream.err; echo rc=$? >> $O/bench_c2_stream.err)
(timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --precision bf16 --no-other-precisions > $O/bench_c3_pipe.json 2>> $O/bench_c3.err)
(timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --precision bf16 --no-other-precisions --no-pipeline > $O/bench_c3_nopipe.json 2>> $O/bench_c3.err)
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o t -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/prof_bench.err
for c in "c5 hist" "c5 dense" "c2 hist" "c4 hist"; do set -- $c
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/sim_$1_$2 -o t -- python3 $R/scripts/sim_only.py $1 $2 20 > $O/sim_$1_$2.log 2>&1
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/simloss_$1_$2 -o t -- python3 $R/scripts/simloss_only.py $1 $2 20 > $O/simloss_$1_$2.log 2>&1
  for pmc in FETCH_SIZE WRITE_SIZE; do
    timeout 120 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $O/pmc_$1_$2_$pmc -o t -- python3 $R/scripts/sim_only.py $1 $2 5 > $O/pmc_$1_$2_$pmc.log 2>&1
  done
done
# (the plain-bf16 GEMM / conv PMC passes of this round were taken by the first run of this script: r03_pmc_bf16_plain_*)
cd $R
export NAFAE_LIB=$R/nafae_amd/csrc/libnafae_hip_exp.so
for c in "c5 hist" "c2 hist" "c5 dense"; do set -- $c
  timeout 120 python3 scripts/simfused_stamps.py $1 $2 > $O/stamps_$1_$2.txt 2>&1
done
NAFAE_SIM_DBG=256 timeout 120 python3 scripts/simfused_trip.py c5 dense > $O/trip_c5_dense.txt 2>&1
unset NAFAE_LIB
cd $R
ls $O | wc -l
