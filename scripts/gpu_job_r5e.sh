set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5e
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_x3 -o t -- python3 $R/bench.py --precision bf16x3 --steps 20 --warmup 3 --no-cpu-baseline --no-other-precisions > $O/bench_x3_under_rocprof.json 2> $O/prof_x3.err
cd $R
python3 - <<'PY'
import csv, json
rows=list(csv.DictReader(open("gpurun_out/r5e/prof_x3/t_kernel_stats.csv")))
for r in rows[:30]:
    print("%-84s calls %5s avg %8.1f us  %5.2f%%" % (r["Name"].replace("(anonymous namespace)::","")[:84], r["Calls"], float(r["AverageNs"])/1e3, float(r["Percentage"])))
d=json.loads(open("gpurun_out/r5e/bench_x3_under_rocprof.json").read().strip().splitlines()[-1])
print("bf16x3 under rocprof", d["value"], d["ms_per_step"])
PY
