#!/bin/bash
# conv-kernel iteration: parity first, then per-layer times and a short bench of the two bf16 modes
mkdir -p gpurun_out/r2g
O=gpurun_out/r2g
(timeout 1500 python -m pytest tests -q -m gpu -x > $O/gpu_all.log 2>&1; echo rc=$? >> $O/gpu_all.log)
grep -E "passed|failed|^FAILED|rc=|Error" $O/gpu_all.log | tail -15
(timeout 300 python scripts/layer_times.py > $O/layers.log 2>&1; echo rc=$? >> $O/layers.log)
cat $O/layers.log
(timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --precision bf16x3 --no-other-precisions > $O/bench_bf16x3.json 2> $O/bench_bf16x3.err; echo rc=$? >> $O/bench_bf16x3.err)
(timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --precision bf16 --no-other-precisions > $O/bench_bf16.json 2> $O/bench_bf16.err; echo rc=$? >> $O/bench_bf16.err)
python - <<'PY'
import json
for n in ("bf16x3","bf16"):
    try:
        b=json.load(open("gpurun_out/r2g/bench_%s.json"%n)); print(n, b["value"], b["ms_per_step"], b.get("stage_ms"))
    except Exception as ex: print(n, "ERR", ex)
PY
