#!/bin/bash
mkdir -p gpurun_out/r2m
O=gpurun_out/r2m
(timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -q -m gpu -x > $O/t.log 2>&1; echo rc=$? >> $O/t.log); tail -3 $O/t.log
(timeout 300 python scripts/layer_times_f32.py > $O/layers_f32.log 2>&1; echo rc=$? >> $O/layers_f32.log); grep -v amdgpu.ids $O/layers_f32.log
(timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-precisions > $O/bench_f32.json 2> $O/bench_f32.err; echo rc=$? >> $O/bench_f32.err)
python3 -c "
import json; b=json.load(open('gpurun_out/r2m/bench_f32.json')); print(b['value'], b['ms_per_step'], b['stage_ms'], b['roofline']['frac'], b['detector'])"
