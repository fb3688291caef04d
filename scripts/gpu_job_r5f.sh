# round 4: PMC passes of the bf16 tile kernels after the in-kernel stream-K fix-up (one counter group per pass):
# bf16x3 (kernels_bf16_only.py) and plain bf16 (kernels_bf16_plain.py) -- fc6-shaped GEMM and a 256->256 @ 56^2 conv each
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5f
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $pmc | cut -d" " -f1)
  timeout 200 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $O/pmcb_$n -o t -- python3 $R/scripts/kernels_bf16_only.py > $O/pmcb_$n.log 2>&1 < /dev/null
  timeout 200 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $O/pmcp_$n -o t -- python3 $R/scripts/kernels_bf16_plain.py > $O/pmcp_$n.log 2>&1 < /dev/null
done
ls $O
