#!/bin/bash
mkdir -p gpurun_out/r2l
O=gpurun_out/r2l
(timeout 1500 python -m pytest tests -q -m gpu -x > $O/gpu_all.log 2>&1; echo rc=$? >> $O/gpu_all.log)
grep -E "passed|failed|^FAILED|rc=|Error" $O/gpu_all.log | tail -6
(timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo rc=$? >> $O/bench.err)
python3 -c "
import json; b=json.load(open('gpurun_out/r2l/bench.json')); print(b['value'], b['ms_per_step'], b['stage_ms'], b['roofline']['frac'], b['detector'])
for m,v in b['modes'].items(): print(m, v['value'], v['ms_per_step'], v['stage_ms'])"
