# round 3, job E: frame-kernel breakdown (experiments build: NAFAE_SIM_DBG bits) + parity re-check
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3e
mkdir -p $O
cd $R
(timeout 1200 python -m pytest tests/test_gpu_simmax.py tests/test_gpu_configs.py -q -m gpu -k "sim" --maxfail=40 > $O/sim_tests.log 2>&1; echo rc=$? >> $O/sim_tests.log)
grep -E "passed|failed|^FAILED|^ERROR|rc=" $O/sim_tests.log | tail -30
cd /tmp; export TMPDIR=/tmp
export NAFAE_LIB=$R/nafae_amd/csrc/libnafae_hip_exp.so
for dbg in 0 1 3 5 9 13 7; do
  NAFAE_SIM_DBG=$dbg timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/dense_dbg$dbg -o t -- python3 $R/scripts/sim_only.py c5 dense 20 > $O/dense_dbg$dbg.log 2>&1
  echo "dbg=$dbg $(grep -E 'sim_frame' $O/dense_dbg$dbg/t_kernel_stats.csv | cut -d, -f2-4,6,7)"
done
NAFAE_SIM_DBG=0 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4dense -o t -- python3 $R/scripts/sim_only.py c4 dense 20 > $O/c4dense.log 2>&1
echo "c4 dense $(grep -E 'sim_frame' $O/c4dense/t_kernel_stats.csv | cut -d, -f2-4,6,7)"
