set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4f
mkdir -p $O
cd $R
(timeout 2400 python -m pytest tests -q -m gpu --maxfail=30 --durations=8 > $O/gpu_all.log 2>&1; echo rc=$? >> $O/gpu_all.log)
grep -E "passed|failed|^FAILED|^ERROR|rc=" $O/gpu_all.log | tail -12
timeout 300 python scripts/jpeg_time.py 0 2>&1 | grep -v amdgpu.ids | tee $O/jpeg_time.txt
timeout 300 python scripts/jpeg_time.py 14 2>&1 | grep -v amdgpu.ids | tee -a $O/jpeg_time.txt
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_jpeg -o t -- python3 $R/scripts/jpeg_time.py 0 > $O/prof_jpeg.log 2>&1
ls $O/prof_jpeg 2>/dev/null | head
