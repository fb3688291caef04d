#!/bin/bash
mkdir -p gpurun_out/r2h
O=gpurun_out/r2h
(timeout 600 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_configs.py -q -m gpu -x > $O/t.log 2>&1; echo rc=$? >> $O/t.log)
grep -E "passed|failed|^FAILED|rc=|Error" $O/t.log | tail -8
python scripts/conv_times.py > $O/times.log 2>&1
for f in nafae_amd/csrc/variants/*.so; do
  NAFAE_LIB=$PWD/$f timeout 200 python scripts/conv_times.py >> $O/times.log 2>&1
done
grep -v amdgpu.ids $O/times.log
