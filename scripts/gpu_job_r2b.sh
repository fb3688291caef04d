set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2b
cd $R
(timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r2b/gpu_all.log 2>&1; echo rc=$? >> gpurun_out/r2b/gpu_all.log)
grep -E "passed|failed|^FAILED|^ERROR|rc=" gpurun_out/r2b/gpu_all.log | tail -12
(timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/r2b/bench.json 2> gpurun_out/r2b/bench.err; echo rc=$? >> gpurun_out/r2b/bench.err)
for mode in bf16x3 f32; do
  timeout 300 python bench.py --steps 20 --warmup 3 --precision $mode --no-other-precisions --no-cpu-baseline > gpurun_out/r2b/bench_${mode}_pipe.json 2>> gpurun_out/r2b/bench.err
  timeout 300 python bench.py --steps 20 --warmup 3 --precision $mode --no-other-precisions --no-cpu-baseline --no-pipeline > gpurun_out/r2b/bench_${mode}_nopipe.json 2>> gpurun_out/r2b/bench.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2b/bench*.json')):
    try:
        d=json.load(open(f)); print(f, d['dtype'], d['value'], d['ms_per_step'], d['config']['step_pipeline'][:12], {k:v for k,v in d['stage_ms'].items() if k in ('base','fc6','fc7','sim_max','vis_ebd','vis_ebd_bwd','word_ebd')})
    except Exception as e: print(f, 'ERR', e)
PY
