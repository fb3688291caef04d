set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3j
mkdir -p $O
cd $R
export NAFAE_LIB=$R/nafae_amd/csrc/libnafae_hip_exp.so
for dbg in 0 32 64 128; do
  echo "== dbg $dbg"
  NAFAE_SIM_DBG=$dbg timeout 120 python3 scripts/simfused_stamps.py c5 dense 2>&1 | grep -E "k-loop|chunk 0|end"
done
