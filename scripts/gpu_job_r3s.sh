set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3s
mkdir -p $O
cd $R
(timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_configs.py tests/test_gpu_exact_dp.py -q -m gpu --maxfail=20 > $O/tests.log 2>&1; echo rc=$? >> $O/tests.log)
grep -E "passed|failed|^FAILED|^ERROR|rc=" $O/tests.log | tail
python3 - <<'P'
import sys; sys.path.insert(0,'.')
import bench, torch
dev = torch.device("cuda:0")
for nm in ("c2","c4","c5"):
    w = bench.WORKLOADS[nm]
    so = bench.sim_loss_only(*w, dev)
    print(nm, "hist  fwd_ms", so["fwd_ms"], "fwd_bwd_ms", so["fwd_bwd_ms"])
c5 = bench.WORKLOADS["c5"]
so = bench.sim_loss_only(*c5, dev, lens=[c5[3]] * c5[0])
print("c5 dense fwd_ms", so["fwd_ms"], "fwd_bwd_ms", so["fwd_bwd_ms"])
P
cd /tmp; export TMPDIR=/tmp
for c in "c5 hist" "c5 dense"; do set -- $c
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/simloss_$1_$2 -o t -- python3 $R/scripts/simloss_only.py $1 $2 20 > $O/simloss_$1_$2.log 2>&1
  grep sim_bwd $O/simloss_$1_$2/t_kernel_stats.csv | cut -d, -f1-4 | cut -c24-60,150-200
done
