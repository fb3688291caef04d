set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4o
mkdir -p $O
cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -Wno-unused-value scripts/micro/wg_placement.hip -o /tmp/wg_placement 2>/dev/null && timeout 120 /tmp/wg_placement | tee $O/wg_placement.txt
timeout 300 python -m pytest tests/test_gpu_jpeg.py -m gpu -q -x 2>&1 | tail -4 | tee $O/tests_jpeg.log
for r in 0 14; do timeout 200 python scripts/jpeg_time.py $r 2>&1 | grep -v amdgpu.ids | tee -a $O/jpeg_time.txt; done
export NAFAE_LIB=$R/nafae_amd/csrc/libnafae_hip_exp.so
timeout 900 python tests/dispatch_worker.py $O/dispatch_table.json > $O/dispatch.log 2>&1; echo "dispatch rc=$?"; tail -3 $O/dispatch.log
export NAFAE_CONV4=0
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/pmc_occ -o t -- python3 $R/scripts/conv_occupancy.py bf16 > $O/pmc_occ.log 2>&1 < /dev/null
cd $R
python3 - <<'PY'
import csv,glob,collections
fs=glob.glob("gpurun_out/r4o/pmc_occ/**/*counter_collection.csv", recursive=True)
rows=list(csv.DictReader(open(fs[0])))
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if "conv3x3_run" not in r["Kernel_Name"]: continue
    key=int(r.get("Grid_Size"))//512
    agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if r["Counter_Name"]=="GRBM_GUI_ACTIVE": agg[key]["ns"].append(float(r["End_Timestamp"])-float(r["Start_Timestamp"]))
for k in sorted(agg):
    c=agg[k]; m={a:sum(b)/len(b) for a,b in c.items()}
    print("%4d workgroups: dur %.1f us  clock %.2f GHz (GRBM_GUI_ACTIVE / 8 XCDs / duration)  MFMA-busy cycles per busy CU / active cycles %.3f" % (k, m["ns"]/1e3, m["GRBM_GUI_ACTIVE"]/8/m["ns"], m["SQ_VALU_MFMA_BUSY_CYCLES"]/(m["GRBM_GUI_ACTIVE"]/8)/(min(k,256)*4)))
PY
