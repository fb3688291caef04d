"""Fold rocprofv3 --pmc CSV passes (one counter group per pass) into profiles/rNN_pmc_counters.json.

usage: python scripts/pmc_summary.py OUT.json KEY=substring[,algorithmic_bytes] ... -- pass1_counter_collection.csv pass2.csv ...
Per kernel whose name contains `substring`: the mean of every counter over its launches (the first launch is dropped as
warm-up when there are several).  traffic_bytes_corrected = (2*FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE
tallies 64 B per 128-B request for 16 B/lane streaming reads (MI355X_MICROARCH.md, HBM section); both are in KiB."""
import csv, json, os, sys
from collections import defaultdict

def main():
    out = sys.argv[1]
    sep = sys.argv.index("--")
    keys = {}
    for kv in sys.argv[2:sep]:
        k, v = kv.split("=", 1)
        sub, _, alg = v.partition(",")
        keys[k] = (sub, int(alg) if alg else None)
    vals = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))   # key -> counter -> dispatch -> value
    for path in sys.argv[sep + 1:]:
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                for k, (sub, _) in keys.items():
                    if sub in row["Kernel_Name"]:
                        vals[k][row["Counter_Name"]][int(row["Dispatch_Id"])] += float(row["Counter_Value"])
    res = json.load(open(out)) if os.path.exists(out) else {}
    for k, (sub, alg) in keys.items():
        e = {}
        for c, d in vals[k].items():
            xs = [d[i] for i in sorted(d)]
            xs = xs[1:] if len(xs) > 1 else xs
            e[c] = sum(xs) / len(xs)
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            e["traffic_bytes_corrected"] = (2 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024
            if alg:
                e["algorithmic_bytes"] = alg
                e["traffic_over_algorithmic"] = e["traffic_bytes_corrected"] / alg
        if "TCC_HIT_sum" in e and "TCC_MISS_sum" in e:
            e["l2_hit_rate"] = e["TCC_HIT_sum"] / (e["TCC_HIT_sum"] + e["TCC_MISS_sum"])
        if "SQ_VALU_MFMA_BUSY_CYCLES" in e and "GRBM_GUI_ACTIVE" in e:
            # SQ_VALU_MFMA_BUSY_CYCLES sums over the 256 CUs x 4 SIMD-quarters; GRBM_GUI_ACTIVE over the 8 XCDs
            e["mfma_busy_frac_of_active_cycles"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / 4 / 256 / (e["GRBM_GUI_ACTIVE"] / 8)
        res[k] = e
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: res[k] for k in keys}, indent=1))

if __name__ == "__main__":
    main()
