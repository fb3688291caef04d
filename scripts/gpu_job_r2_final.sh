set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2z
mkdir -p $O
cd $R
(timeout 1500 python -m pytest tests -q -m gpu > $O/gpu_all.log 2>&1; echo rc=$? >> $O/gpu_all.log)
grep -E "passed|failed|^FAILED|^ERROR|rc=" $O/gpu_all.log | tail -8
(timeout 600 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err; echo rc=$? >> $O/bench.err)
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o t -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/prof_bench.err
for c in "c5 hist" "c5 dense" "c2 hist"; do set -- $c
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/sim_$1_$2 -o t -- python3 $R/scripts/sim_only.py $1 $2 20 > $O/sim_$1_$2.log 2>&1
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/simloss_$1_$2 -o t -- python3 $R/scripts/simloss_only.py $1 $2 20 > $O/simloss_$1_$2.log 2>&1
  for pmc in FETCH_SIZE WRITE_SIZE; do
    timeout 120 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $O/pmc_$1_$2_$pmc -o t -- python3 $R/scripts/sim_only.py $1 $2 5 > $O/pmc_$1_$2_$pmc.log 2>&1
  done
done
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $pmc | cut -d" " -f1)
  timeout 200 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $O/pmck_$n -o t -- python3 $R/scripts/kernels_only.py 3 > $O/pmck_$n.log 2>&1
done
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $pmc | cut -d" " -f1)
  timeout 200 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $O/pmcb_$n -o t -- python3 $R/scripts/kernels_bf16_only.py > $O/pmcb_$n.log 2>&1 < /dev/null
done
cd $R
(timeout 300 python scripts/layer_times_f32.py > $O/layers_f32.log 2>&1); (timeout 300 python scripts/layer_times.py > $O/layers_bf16x3.log 2>&1)
ls $O | wc -l
