# round 4, first job: the whole GPU suite (new: RCCL world-size-1, accuracy delta) + tail-stream priority A/B in bf16 / f32
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4a
mkdir -p $O
cd $R
(timeout 2400 python -m pytest tests -q -m gpu --maxfail=30 --durations=8 > $O/gpu_all.log 2>&1; echo rc=$? >> $O/gpu_all.log)
grep -E "passed|failed|^FAILED|^ERROR|rc=" $O/gpu_all.log | tail -12
grep -E "^\[C[245]" $O/gpu_all.log > $O/gpu_tests_summary.txt
grep -E "accuracy" $O/gpu_tests_summary.txt
for p in bf16 f32; do
  for v in "" "--no-tail-priority" "--no-pipeline"; do
    n=$(echo "$p$v" | tr -d ' ' | tr -- '-' '_')
    (timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --precision $p --no-other-precisions $v > $O/bench_$n.json 2> $O/bench_$n.err; echo rc=$? >> $O/bench_$n.err)
    python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$n.json").read().strip().splitlines()[-1])
    st=d.get("stage_ms",{})
    print("$n", d["value"], d["ms_per_step"], {k:st[k] for k in ("sim_max","loss_tail","vis_ebd","vis_ebd_bwd","sim_bwd","word_ebd") if k in st})
except Exception as e:
    print("$n failed", e)
PY
  done
done
(timeout 300 python bench.py --gpus 1 --force-dist --steps 10 --warmup 2 --no-cpu-baseline --no-other-precisions > $O/bench_force_dist.json 2> $O/bench_force_dist.err; echo rc=$? >> $O/bench_force_dist.err)
python -c "import json;d=json.loads(open('$O/bench_force_dist.json').read().strip().splitlines()[-1]);print('force-dist',d['value'],d['config']['collective_backend'])"
