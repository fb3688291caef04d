# round 3, job B: whole -m gpu suite, all failures listed
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3b
mkdir -p $O
cd $R
(timeout 3000 python -m pytest tests -q -m gpu --maxfail=30 --durations=15 > $O/gpu_all.log 2>&1; echo rc=$? >> $O/gpu_all.log)
grep -E "passed|failed|^FAILED|^ERROR|rc=" $O/gpu_all.log | tail -40
grep -E "^\[C[245]" $O/gpu_all.log > $O/gpu_tests_summary.txt
