set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3n
mkdir -p $O
cd $R
(timeout 1500 python -m pytest tests/test_gpu_simmax.py tests/test_gpu_stress.py tests/test_gpu_configs.py tests/test_gpu_exact_dp.py -q -m gpu --maxfail=40 > $O/sim_tests.log 2>&1; echo rc=$? >> $O/sim_tests.log)
grep -E "passed|failed|^FAILED|^ERROR|rc=" $O/sim_tests.log | tail -30
cd /tmp; export TMPDIR=/tmp
for c in "c5 hist" "c2 hist" "c4 hist"; do set -- $c
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/sim_$1_$2 -o t -- python3 $R/scripts/sim_only.py $1 $2 20 > $O/sim_$1_$2.log 2>&1
  grep -E "sim_" $O/sim_$1_$2/t_kernel_stats.csv | cut -d, -f1-4,6,7 | cut -c1-50,140-260
done
cd $R
export NAFAE_LIB=$R/nafae_amd/csrc/libnafae_hip_exp.so
for c in "c5 hist" "c2 hist"; do set -- $c
  timeout 120 python3 scripts/simfused_stamps.py $1 $2 2>&1 | grep -v amdgpu.ids
done
python3 - <<'P'
import sys; sys.path.insert(0,'.')
import bench, torch
dev = torch.device("cuda:0")
for nm in ("c2","c4","c5"):
    w = bench.WORKLOADS[nm]
    so = bench.sim_loss_only(*w, dev)
    print(nm, "hist fwd_ms", so["fwd_ms"], "fwd_bwd_ms", so["fwd_bwd_ms"], "hbm_frac", so["fwd_hbm_frac"])
P
