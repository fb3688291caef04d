set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3p
mkdir -p $O
cd $R
export NAFAE_LIB=$R/nafae_amd/csrc/libnafae_hip_exp.so
for n in 2 3; do
  echo "== NSET $n"
  for c in "c5 dense" "c4 dense"; do set -- $c
    NAFAE_SIM_NSET=$n timeout 120 python3 scripts/simfused_stamps.py $1 $2 2>&1 | grep -v amdgpu.ids
  done
done
(NAFAE_SIM_NSET=3 timeout 900 python -m pytest tests/test_gpu_simmax.py tests/test_gpu_configs.py -q -m gpu -k "sim or dense or live" --maxfail=20 > $O/sim_tests_nset3.log 2>&1; echo rc=$? >> $O/sim_tests_nset3.log)
grep -E "passed|failed|^FAILED|^ERROR|rc=" $O/sim_tests_nset3.log | tail
cd /tmp; export TMPDIR=/tmp
for n in 2 3; do
  NAFAE_SIM_NSET=$n timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/sim_c5_dense_$n -o t -- python3 $R/scripts/sim_only.py c5 dense 20 > $O/sim_c5_dense_$n.log 2>&1
  grep -E "sim_" $O/sim_c5_dense_$n/t_kernel_stats.csv | cut -d, -f1-4,6,7 | cut -c24-70,140-260
done
