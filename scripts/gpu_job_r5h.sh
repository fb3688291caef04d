set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5h
mkdir -p $O
cd $R
for rep in 1 2; do
for arm in memset zeroonce; do
  if [ $arm = memset ]; then export NAFAE_LIB=$R/nafae_amd/csrc/variants/libnafae_hip_memset.so; else unset NAFAE_LIB; fi
  timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>$O/bench.err | tail -1 > $O/bench_${arm}_$rep.json
  python - <<PY
import json
d=json.load(open("gpurun_out/r5h/bench_${arm}_$rep.json"))
print("$arm", d.get("value"), d.get("ms_per_step"), {k:v["value"] for k,v in d.get("modes",{}).items()})
PY
done
done
