set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4p
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_jpeg.py -m gpu -q -x 2>&1 | tail -12 | tee $O/tests_jpeg.log
for r in 0 14; do timeout 200 python scripts/jpeg_time.py $r 2>&1 | grep -v amdgpu.ids | tee -a $O/jpeg_time.txt; done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/jpeg_prof -o t -- python3 $R/scripts/jpeg_time.py 0 > $O/jpeg_prof.log 2>&1 < /dev/null
cd $R
grep -h "jpeg" $O/jpeg_prof/*kernel_stats.csv 2>/dev/null | head
(timeout 2400 python -m pytest tests -q -m gpu --maxfail=10 --durations=5 > $O/gpu_all.log 2>&1; echo rc=$? >> $O/gpu_all.log)
grep -E "passed|failed|^FAILED|^ERROR|rc=" $O/gpu_all.log | tail -12
