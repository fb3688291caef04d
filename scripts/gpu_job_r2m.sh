#!/bin/bash
mkdir -p gpurun_out/r2m
O=gpurun_out/r2m
(timeout 300 python scripts/layer_times_f32.py > $O/layers_f32.log 2>&1; echo rc=$? >> $O/layers_f32.log); grep -v amdgpu.ids $O/layers_f32.log | head -3
(timeout 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k conv > $O/t.log 2>&1; echo rc=$? >> $O/t.log); tail -3 $O/t.log
