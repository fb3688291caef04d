# two ranks sharing ONE GPU (bench.py's test mode, gloo): the step time of the three arithmetic modes, repeated -- the stream-K
# convs' in-kernel hand-off must not crawl when another process's kernels hold part of the CUs
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/shared
mkdir -p $O
cd $R
(timeout 2400 python -m pytest tests -q -m gpu --maxfail=10 > $O/gpu_all.log 2>&1; echo rc=$? >> $O/gpu_all.log)
grep -E "passed|failed|^FAILED|^ERROR|rc=" $O/gpu_all.log | tail -6
for rep in 1 2 3; do
  for prec in f32 bf16x3 bf16; do
    timeout 600 python bench.py --gpus 2 --test-shared-gpu --precision $prec --steps 5 --warmup 2 --no-cpu-baseline --no-other-precisions 2>$O/err.txt | grep -v Gloo | tail -1 > $O/o.json
    python - <<PY
import json
d=json.load(open("gpurun_out/shared/o.json"))
print("shared 2 ranks $prec", d.get("value"), d.get("ms_per_step"))
PY
  done
done
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>$O/bench.err | tail -1 > $O/bench_f32.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/shared/bench_f32.json"))
print("f32", d.get("value"), d.get("ms_per_step"), {k:v["value"] for k,v in d.get("modes",{}).items()})
PY
done
