set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4m
mkdir -p $O
cd $R
timeout 300 python scripts/debug/conv4_debug.py 2>&1 | grep -v amdgpu.ids > $O/conv4_debug.txt
timeout 600 python -m pytest tests/test_gpu_simmax.py tests/test_gpu_simplanes.py tests/test_gpu_model.py -m gpu -q -x 2>&1 | tail -8 | tee $O/tests_sim.log
