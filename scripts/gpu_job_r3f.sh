# round 3, job F: whole gpu suite (ingest, run entry, loss-tail per segment, tail kernels), then sim breakdowns
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3f
mkdir -p $O
cd $R
(timeout 2400 python -m pytest tests -q -m gpu --maxfail=30 --durations=10 > $O/gpu_all.log 2>&1; echo rc=$? >> $O/gpu_all.log)
grep -E "passed|failed|^FAILED|^ERROR|rc=" $O/gpu_all.log | tail -30
grep -E "^\[C[245]" $O/gpu_all.log > $O/gpu_tests_summary.txt
cd /tmp; export TMPDIR=/tmp
export NAFAE_LIB=$R/nafae_amd/csrc/libnafae_hip_exp.so
for dbg in 0 1 17; do
  NAFAE_SIM_DBG=$dbg timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/dense_dbg$dbg -o t -- python3 $R/scripts/sim_only.py c5 dense 20 > $O/dense_dbg$dbg.log 2>&1
done
for dbg in 0 1; do
  NAFAE_SIM_DBG=$dbg timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/hist_dbg$dbg -o t -- python3 $R/scripts/sim_only.py c5 hist 20 > $O/hist_dbg$dbg.log 2>&1
done
for c in "c5 hist" "c5 dense"; do set -- $c
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/simloss_$1_$2 -o t -- python3 $R/scripts/simloss_only.py $1 $2 20 > $O/simloss_$1_$2.log 2>&1
  NAFAE_LOSS_SEG=1 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/simloss_seg_$1_$2 -o t -- python3 $R/scripts/simloss_only.py $1 $2 20 > $O/simloss_seg_$1_$2.log 2>&1
done
ls $O | wc -l
