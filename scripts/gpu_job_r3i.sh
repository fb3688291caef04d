set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3i
mkdir -p $O
cd $R
export NAFAE_LIB=$R/nafae_amd/csrc/libnafae_hip_exp.so
for c in "c5 hist" "c5 dense" "c2 hist"; do set -- $c
  timeout 120 python3 scripts/simfused_stamps.py $1 $2 > $O/stamps_$1_$2.log 2>&1
  cat $O/stamps_$1_$2.log
done
