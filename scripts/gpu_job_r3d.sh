# round 3, job D: similarity kernels again (load order in the few-column kernel, workgroup-wide slow path in the frame kernel) + breakdown
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3d
mkdir -p $O
cd $R
(timeout 1200 python -m pytest tests/test_gpu_simmax.py tests/test_gpu_configs.py tests/test_gpu_ops.py -q -m gpu -k "sim or tail_kernels" --maxfail=40 > $O/sim_tests.log 2>&1; echo rc=$? >> $O/sim_tests.log)
grep -E "passed|failed|^FAILED|^ERROR|rc=" $O/sim_tests.log | tail -30
cd /tmp; export TMPDIR=/tmp
for c in "c5 hist" "c5 dense" "c2 hist" "c4 dense"; do set -- $c
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/sim_$1_$2 -o t -- python3 $R/scripts/sim_only.py $1 $2 20 > $O/sim_$1_$2.log 2>&1
  grep -E "sim_" $O/sim_$1_$2/t_kernel_stats.csv | cut -d, -f1-4,6,7 | cut -c1-60,150-260
done
export NAFAE_LIB=$R/nafae_amd/csrc/libnafae_hip_exp.so
NAFAE_SIM_DBG=1 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/sim_c5_dense_dbg1 -o t -- python3 $R/scripts/sim_only.py c5 dense 20 > $O/sim_c5_dense_dbg1.log 2>&1
grep -E "sim_" $O/sim_c5_dense_dbg1/t_kernel_stats.csv | cut -d, -f1-4,6,7 | cut -c1-60,150-260
