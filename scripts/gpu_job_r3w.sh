set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3w
mkdir -p $O
cd $R
(timeout 2400 python -m pytest tests -q -m gpu --maxfail=30 > $O/gpu_all.log 2>&1; echo rc=$? >> $O/gpu_all.log)
grep -E "passed|failed|^FAILED|^ERROR|rc=" $O/gpu_all.log | tail -20
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python scripts/sim_sweep_live.py c5 2>&1 | grep -v amdgpu
