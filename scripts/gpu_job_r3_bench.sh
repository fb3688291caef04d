# round 3: the bench lines only (scripts/gpu_job_r3_final.sh without the test suite, the similarity traces and the PMC passes)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3b
mkdir -p $O
cd $R
(timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench_c2.json 2> $O/bench_c2.err; echo rc=$? >> $O/bench_c2.err)
for w in c4 c5; do
  (timeout 600 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err; echo rc=$? >> $O/bench_$w.err)
done
(timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --stream-input > $O/bench_c2_stream.json 2> $O/bench_c2_stream.err; echo rc=$? >> $O/bench_c2_stream.err)
(timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --precision bf16 --no-other-precisions > $O/bench_c3_pipe.json 2>> $O/bench_c3.err)
(timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --precision bf16 --no-other-precisions --no-pipeline > $O/bench_c3_nopipe.json 2>> $O/bench_c3.err)
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o t -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/prof_bench.err
cd $R; ls $O
