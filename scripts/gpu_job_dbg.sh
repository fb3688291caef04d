set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/dbg
mkdir -p $O
cd $R
timeout -s ABRT 150 python -X faulthandler bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/out.txt 2> $O/err.txt; echo "rc=$?"
tail -40 $O/err.txt | cut -c1-200
tail -2 $O/out.txt | cut -c1-300
