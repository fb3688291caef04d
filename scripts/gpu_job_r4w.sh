set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4w
mkdir -p $O
cd $R
NAFAE_LIB=$R/nafae_amd/csrc/libnafae_hip_exp.so timeout 900 python tests/dispatch_worker.py $O/dispatch_table.json > $O/dispatch.log 2>&1; echo "dispatch rc=$?"; tail -2 $O/dispatch.log
(timeout 2400 python -m pytest tests -q -m gpu --maxfail=10 --durations=5 --deselect tests/test_gpu_dispatch.py > $O/gpu_all.log 2>&1; echo rc=$? >> $O/gpu_all.log)
grep -E "passed|failed|^FAILED|^ERROR|rc=" $O/gpu_all.log | tail -12
