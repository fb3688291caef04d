set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4q
mkdir -p $O
cd $R
export NAFAE_LIB=$R/nafae_amd/csrc/libnafae_hip_exp.so
NAFAE_PAIR_DEEP=0 timeout 300 python tests/exp_arm_worker.py pair /tmp/pair0.pt
NAFAE_PAIR_DEEP=1 timeout 300 python tests/exp_arm_worker.py pair /tmp/pair1.pt
python - <<'PY' 2>&1 | tee $O/pair_deep_check.txt
import torch
a=torch.load("/tmp/pair0.pt"); b=torch.load("/tmp/pair1.pt")
for c in a:
    e=max(float((x-y).abs().max()) for x,y in zip(a[c],b[c]))
    print(c, "max abs diff deep vs default: %.3g (scale %.3g)  bit-identical: %s" % (e, float(a[c][0].abs().max()), all(torch.equal(x,y) for x,y in zip(a[c],b[c]))))
PY
for k in 0 1 0 1; do
  echo "== NAFAE_PAIR_DEEP=$k bf16" | tee -a $O/layers.txt
  NAFAE_PAIR_DEEP=$k timeout 300 python scripts/layer_times.py bf16 2>&1 | grep -v amdgpu.ids | tee -a $O/layers.txt
done
