set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3q
mkdir -p $O
cd $R
(timeout 900 python -m pytest tests/test_gpu_simmax.py tests/test_gpu_configs.py tests/test_gpu_stress.py -q -m gpu --maxfail=20 > $O/sim_tests.log 2>&1; echo rc=$? >> $O/sim_tests.log)
grep -E "passed|failed|^FAILED|^ERROR|rc=" $O/sim_tests.log | tail
export NAFAE_LIB=$R/nafae_amd/csrc/libnafae_hip_exp.so
for c in "c5 dense" "c4 dense"; do set -- $c
  timeout 120 python3 scripts/simfused_stamps.py $1 $2 2>&1 | grep -v amdgpu.ids
done
for d in 256 260; do
  NAFAE_SIM_DBG=$d timeout 120 python3 scripts/simfused_trip.py c5 dense 2>&1 | grep -v amdgpu.ids
done
unset NAFAE_LIB
cd /tmp; export TMPDIR=/tmp
for c in "c5 dense" "c4 dense"; do set -- $c
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/sim_$1_$2 -o t -- python3 $R/scripts/sim_only.py $1 $2 20 > $O/sim_$1_$2.log 2>&1
  grep -E "sim_" $O/sim_$1_$2/t_kernel_stats.csv | cut -d, -f1-4,6,7 | cut -c24-70,140-260
done
