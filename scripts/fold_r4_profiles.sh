# copy the judged summaries of gpurun_out/<dir> into profiles/ under round-4 names and fold the PMC passes
#   bash scripts/fold_r4_profiles.sh r4p
set -eu
O=gpurun_out/$1
P=profiles
cp $O/bench_c2.json $P/r04_bench_c2.json
cp $O/bench_c4.json $P/r04_bench_c4.json
cp $O/bench_c5.json $P/r04_bench_c5.json
cp $O/bench_c2_force_dist.json $P/r04_bench_c2_force_dist_nccl_world1.json
cp $O/bench_under_rocprof.json $P/r04_bench_c2_under_rocprof.json
cp $O/prof_bench/t_kernel_stats.csv $P/r04_bench_c2_kernel_stats.csv
for c in c5_hist c5_dense c2_hist c4_hist; do
  cp $O/sim_$c/t_kernel_stats.csv $P/r04_sim_${c}_kernel_stats.csv
  cp $O/simloss_$c/t_kernel_stats.csv $P/r04_simloss_${c}_kernel_stats.csv
  cp $O/pmc_${c}_FETCH_SIZE/t_counter_collection.csv $P/r04_pmc_sim_${c}_FETCH_SIZE.csv
  cp $O/pmc_${c}_WRITE_SIZE/t_counter_collection.csv $P/r04_pmc_sim_${c}_WRITE_SIZE.csv
done
rm -f $P/r04_pmc_counters.json
python scripts/pmc_fold.py $P/r04_pmc_counters.json sim_c5_hist "sim_" 40763392 $P/r04_pmc_sim_c5_hist_FETCH_SIZE.csv $P/r04_pmc_sim_c5_hist_WRITE_SIZE.csv
python scripts/pmc_fold.py $P/r04_pmc_counters.json sim_c5_dense "sim_" 40763392 $P/r04_pmc_sim_c5_dense_FETCH_SIZE.csv $P/r04_pmc_sim_c5_dense_WRITE_SIZE.csv
python scripts/pmc_fold.py $P/r04_pmc_counters.json sim_c2_hist "sim_" 17137664 $P/r04_pmc_sim_c2_hist_FETCH_SIZE.csv $P/r04_pmc_sim_c2_hist_WRITE_SIZE.csv
python scripts/pmc_fold.py $P/r04_pmc_counters.json sim_c4_hist "sim_" 34275328 $P/r04_pmc_sim_c4_hist_FETCH_SIZE.csv $P/r04_pmc_sim_c4_hist_WRITE_SIZE.csv
