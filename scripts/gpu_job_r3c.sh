# round 3, job C: the third-generation similarity kernels: parity tests, then kernel traces at C5 / C2 / C4
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c
mkdir -p $O
cd $R
(timeout 1200 python -m pytest tests/test_gpu_simmax.py tests/test_gpu_configs.py -q -m gpu -k "sim" --maxfail=40 > $O/sim_tests.log 2>&1; echo rc=$? >> $O/sim_tests.log)
grep -E "passed|failed|^FAILED|^ERROR|rc=" $O/sim_tests.log | tail -30
cd /tmp; export TMPDIR=/tmp
for c in "c5 hist" "c5 dense" "c2 hist" "c4 hist" "c4 dense"; do set -- $c
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/sim_$1_$2 -o t -- python3 $R/scripts/sim_only.py $1 $2 20 > $O/sim_$1_$2.log 2>&1
  grep -E "sim_|Name" $O/sim_$1_$2/t_kernel_stats.csv | cut -d, -f1-4,6,7 | cut -c1-200
done
