#!/bin/bash
# clocks under the conv kernel variants: GRBM_GUI_ACTIVE (GPU-clock cycles, summed over the 8 XCDs) against the dispatch duration
mkdir -p gpurun_out/r2i
O=$PWD/gpurun_out/r2i
R=$PWD
cd /tmp && export TMPDIR=/tmp
for v in default v8 v16 v15; do
  if [ $v = default ]; then unset NAFAE_LIB; else export NAFAE_LIB=$R/nafae_amd/csrc/variants/libnafae_hip_$v.so; fi
  timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/pmc_$v -o t -- python3 $R/scripts/conv_times.py c32 c42 > $O/pmc_$v.log 2>&1 < /dev/null
  NAFAE_LIB=$NAFAE_LIB timeout 100 python3 $R/scripts/conv_times.py c32 c42 2>&1 | grep -v amdgpu.ids
done
cd $R
python3 - <<'PY'
import csv,glob,collections
for v in ("default","v8","v16","v15"):
    fs=glob.glob("gpurun_out/r2i/pmc_%s/**/*counter_collection.csv"%v, recursive=True)
    if not fs: print(v,"no csv"); continue
    rows=list(csv.DictReader(open(fs[0])))
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        if "conv3x3_run" not in r["Kernel_Name"]: continue
        key=(r["Kernel_Name"][:60], r.get("Grid_Size"))
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if "Start_Timestamp" in r: agg[key]["ns"].append(float(r["End_Timestamp"])-float(r["Start_Timestamp"]))
    for k,c in agg.items():
        m={a:sum(b)/len(b) for a,b in c.items()}
        line="%s %s n=%d"%(v,k[0][30:],len(c["GRBM_GUI_ACTIVE"]))
        if "ns" in m: line+=" dur %.1f us clock %.2f GHz"%(m["ns"]/1e3, m["GRBM_GUI_ACTIVE"]/8/m["ns"])
        line+=" mfma_busy/active %.3f"%(m["SQ_VALU_MFMA_BUSY_CYCLES"]/m["GRBM_GUI_ACTIVE"]/ (256*4/8) ) if "SQ_VALU_MFMA_BUSY_CYCLES" in m else ""
        print(line, {a:round(b) for a,b in m.items()})
PY
