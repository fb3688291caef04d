#!/bin/bash
mkdir -p gpurun_out/r2k
O=gpurun_out/r2k
(timeout 300 python scripts/layer_times_f32.py > $O/layers_f32.log 2>&1; echo rc=$? >> $O/layers_f32.log); grep -v amdgpu.ids $O/layers_f32.log
(NAFAE_LIB=$PWD/nafae_amd/csrc/variants/libnafae_hip_skw2.so timeout 300 python scripts/layer_times_f32.py > $O/layers_f32_w2.log 2>&1; echo rc=$? >> $O/layers_f32_w2.log); grep -v amdgpu.ids $O/layers_f32_w2.log
