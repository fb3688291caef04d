# round 3, job A: whole -m gpu suite (C4 / C5 at size, stress sweeps, tightened bars), bench lines for C2 / C4 / C5, kernel trace of C4
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3a
mkdir -p $O
cd $R
(timeout 2400 python -m pytest tests -q -m gpu -x --durations=15 > $O/gpu_all.log 2>&1; echo rc=$? >> $O/gpu_all.log)
grep -E "passed|failed|^FAILED|^ERROR|rc=" $O/gpu_all.log | tail -8
grep -E "^\[C[245]" $O/gpu_all.log > $O/gpu_tests_summary.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
for w in c4 c5; do
  (timeout 600 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err; echo rc=$? >> $O/bench_$w.err)
done
(timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench_c2.json 2> $O/bench_c2.err; echo rc=$? >> $O/bench_c2.err)
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4 -o t -- python3 $R/bench.py --workload c4 --steps 10 --warmup 3 --no-cpu-baseline --no-other-precisions > $O/bench_c4_under_rocprof.json 2> $O/prof_c4.err
cd $R
ls $O | wc -l
