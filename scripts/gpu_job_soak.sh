# soak: long pipelined runs in the three arithmetic modes (the stream-K convs' in-kernel hand-off under the tail stream's kernels,
# streamed input): must finish, and the per-step time must not drift
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/soak
mkdir -p $O
cd $R
for prec in f32 bf16x3 bf16; do
  timeout 600 python bench.py --precision $prec --steps 400 --warmup 5 --stream-input --no-cpu-baseline --no-other-precisions 2>$O/$prec.err | tail -1 > $O/$prec.json; echo "rc=$?"
  python - <<PY
import json
d=json.load(open("gpurun_out/soak/$prec.json"))
print("$prec", d.get("value"), d.get("ms_per_step"), "loss", d.get("loss"))
PY
done
timeout 600 python bench.py --gpus 2 --test-shared-gpu --steps 20 --warmup 3 --no-cpu-baseline --no-other-precisions 2>$O/shared.err | tail -1 > $O/shared2.json; echo "rc=$?"; head -c 300 $O/shared2.json; echo
