# round 3: fresh PMC passes of the f32 and bf16x3 tile kernels (one counter group per pass), the new width test, one more bench line
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3t
mkdir -p $O
cd $R
(timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "dvsa" > $O/tests.log 2>&1; echo rc=$? >> $O/tests.log)
grep -E "passed|failed|^FAILED|^ERROR|rc=" $O/tests.log | tail
(timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench_c2.json 2> $O/bench_c2.err; echo rc=$? >> $O/bench_c2.err)
cd /tmp; export TMPDIR=/tmp
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $pmc | cut -d" " -f1)
  timeout 200 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $O/pmck_$n -o t -- python3 $R/scripts/kernels_only.py 3 > $O/pmck_$n.log 2>&1 < /dev/null
  timeout 200 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $O/pmcb_$n -o t -- python3 $R/scripts/kernels_bf16_only.py > $O/pmcb_$n.log 2>&1 < /dev/null
done
cd $R
ls $O
