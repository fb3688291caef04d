#!/bin/bash
# Variant builds of the Winograd conv kernel: scripts/wino_variants.sh name=FLAGS ...   (e.g. nodma=-DWN_DBG=3)
# -> nafae_amd/csrc/variants/libnafae_hip_<name>.so (the production objects of the other translation units + a variant wino.o);
# select with NAFAE_LIB=<path>.  Timing experiments only: WN_DBG != 0 computes garbage.
set -e
cd "$(dirname "$0")/../nafae_amd/csrc"
mkdir -p variants
OBJS=$(ls *.o | grep -v "_exp.o" | grep -v "^wino.o")
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  /opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -std=c++17 -Wall -Wno-unused-function $flags -c wino.hip -o variants/wino_$name.o &
done
wait
for spec in "$@"; do
  name=${spec%%=*}
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o variants/libnafae_hip_$name.so $OBJS variants/wino_$name.o
  echo variants/libnafae_hip_$name.so
done
