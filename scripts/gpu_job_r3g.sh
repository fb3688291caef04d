# round 3, job G: sim / loss chain after the few-kernel window, merge trim, dV frame kernel with 16 waves
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3g
mkdir -p $O
cd $R
(timeout 2400 python -m pytest tests/test_gpu_simmax.py tests/test_gpu_configs.py tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_widened.py -q -m gpu --maxfail=30 > $O/tests.log 2>&1; echo rc=$? >> $O/tests.log)
grep -E "passed|failed|^FAILED|^ERROR|rc=" $O/tests.log | tail -30
cd /tmp; export TMPDIR=/tmp
for c in "c5 hist" "c5 dense" "c2 hist"; do set -- $c
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/simloss_$1_$2 -o t -- python3 $R/scripts/simloss_only.py $1 $2 20 > $O/simloss_$1_$2.log 2>&1
done
export NAFAE_LIB=$R/nafae_amd/csrc/libnafae_hip_exp.so
for dbg in 16; do
  NAFAE_SIM_DBG=$dbg timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/dense_dbg$dbg -o t -- python3 $R/scripts/sim_only.py c5 dense 20 > $O/dense_dbg$dbg.log 2>&1
done
