#!/bin/bash
# Builds library variants for kernel A/B timing: each argument is  name=FLAGS  (FLAGS = extra hipcc flags for gemm_bf16.hip, e.g.
# v8=-DNAFAE_CONV_EXP=8  or  builtin=-DNAFAE_ASM_DMA=0); the other objects are the ones nafae_amd.build made.  Output:
# nafae_amd/csrc/variants/libnafae_hip_<name>.so -- select with NAFAE_LIB=<path>.  NAFAE_CONV_EXP != 0 gives wrong results (timing only).
set -e
cd "$(dirname "$0")/.."
python -m nafae_amd.build > /dev/null
mkdir -p nafae_amd/csrc/variants
rm -f nafae_amd/csrc/variants/*.so
for a in "$@"; do
  n=${a%%=*}; f=${a#*=}
  /opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -std=c++17 $f \
      -c nafae_amd/csrc/gemm_bf16.hip -o nafae_amd/csrc/variants/gemm_bf16_$n.o 2> /dev/null &
done
wait
for a in "$@"; do
  n=${a%%=*}
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o nafae_amd/csrc/variants/libnafae_hip_$n.so nafae_amd/csrc/variants/gemm_bf16_$n.o \
      nafae_amd/csrc/gemm.o nafae_amd/csrc/proposal.o nafae_amd/csrc/simloss.o nafae_amd/csrc/simmax.o nafae_amd/csrc/simfused.o \
      nafae_amd/csrc/simplanes.o nafae_amd/csrc/jpeg.o
  rm nafae_amd/csrc/variants/gemm_bf16_$n.o
done
ls nafae_amd/csrc/variants/
