#!/bin/bash
# round 5: Winograd kernel timing experiments over variant builds (scripts/wino_variants.sh; WN_DBG != 0 computes garbage by design)
mkdir -p gpurun_out/wdbg
rm -f gpurun_out/wdbg/times.log
for v in ${VARIANTS:-d0 d3 d15 d31 d47 d79 d127}; do
  echo "== variant $v" >> gpurun_out/wdbg/times.log
  for l in ${LAYERS:-conv4_2 conv1_2}; do
    NAFAE_LIB=nafae_amd/csrc/variants/libnafae_hip_$v.so ONLY=$l timeout 120 python scripts/layer_times_wino.py 2>&1 | grep -v "amdgpu.ids\|^sum" | sed 's/ direct-equivalent//; s/| direct.*//' >> gpurun_out/wdbg/times.log
  done
done
cat gpurun_out/wdbg/times.log
