# round 4: the whole GPU suite + smoke
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4t
mkdir -p $O
cd $R
(timeout 2400 python -m pytest tests -q -m gpu --maxfail=30 --durations=8 > $O/gpu_all.log 2>&1; echo rc=$? >> $O/gpu_all.log)
grep -E "passed|failed|^FAILED|^ERROR|rc=" $O/gpu_all.log | tail -12
grep -E "accuracy|^\[C[245]" $O/gpu_all.log > $O/gpu_tests_summary.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
