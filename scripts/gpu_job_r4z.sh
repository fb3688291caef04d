set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4z
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_widened.py tests/test_gpu_ops.py tests/test_gpu_jpeg.py -m gpu -q -x 2>&1 | tail -4 | tee $O/tests.log
for prec in bf16 bf16x3; do timeout 200 python scripts/layer_times.py $prec 2>&1 | grep "conv1_1" | tee -a $O/conv1.txt; done
timeout 200 python scripts/layer_times_f32.py 2>&1 | grep "conv1_1" | tee -a $O/conv1.txt
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3 -o t -- python3 $R/bench.py --precision bf16 --steps 20 --warmup 3 --no-cpu-baseline --no-other-precisions > $O/bench_c3_under_rocprof.json 2> $O/prof_c3.err
cd $R
python3 - <<'PY'
import csv, json
rows=list(csv.DictReader(open("gpurun_out/r4z/prof_c3/t_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:28]:
    print("%-84s calls %5s avg %8.1f us  %5.2f%%" % (r["Name"].replace("(anonymous namespace)::","")[:84], r["Calls"], float(r["AverageNs"])/1e3, float(r["Percentage"])))
d=json.loads(open("gpurun_out/r4z/bench_c3_under_rocprof.json").read().strip().splitlines()[-1])
print("C3 under rocprof", d["value"], d["ms_per_step"])
PY
