set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3g
mkdir -p $O
cd $R
(timeout 2400 python -m pytest tests -q -m gpu --maxfail=30 > $O/gpu_all.log 2>&1; echo rc=$? >> $O/gpu_all.log)
grep -E "passed|failed|^FAILED|^ERROR|rc=" $O/gpu_all.log | tail -20
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
for i in 1 2; do
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --precision bf16x3 --no-other-precisions 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16x3', d['value'], d['ms_per_step'], d['stage_ms'])"
NAFAE_LIB=$R/nafae_amd/csrc/libnafae_hip_exp.so NAFAE_GEMM4=0 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --precision bf16x3 --no-other-precisions 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16x3 8-wave gemm (exp build)', d['value'], d['ms_per_step'], d['stage_ms'])"
NAFAE_LIB=$R/nafae_amd/csrc/libnafae_hip_exp.so python bench.py --steps 20 --warmup 3 --no-cpu-baseline --precision bf16x3 --no-other-precisions 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16x3 4-wave gemm (exp build)', d['value'], d['ms_per_step'], d['stage_ms'])"
done
