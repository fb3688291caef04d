"""Fold rocprofv3 --pmc passes of ONE program into a per-call summary: for every kernel name matching `substr`, the mean counter
value per launch (first launch dropped), summed over the kernels of a call.  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE tallies 64 B per 128-B request of a 16-B-per-lane streaming read (MI355X_MICROARCH.md, HBM section), so
traffic_bytes_corrected = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.

    python scripts/pmc_fold.py OUT.json KEY "substr" ALGORITHMIC_BYTES FETCH_counter_collection.csv WRITE_counter_collection.csv
"""
import csv, json, os, sys
from collections import defaultdict


def per_kernel_means(path, substr):
    vals = defaultdict(lambda: defaultdict(list))          # counter -> kernel -> [per dispatch]
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if substr in row["Kernel_Name"]:
                name = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
                vals[row["Counter_Name"]][name.split("(")[0]].append(float(row["Counter_Value"]))
    out = {}
    for c, ks in vals.items():
        out[c] = {k: (sum(v[1:]) / len(v[1:]) if len(v) > 1 else v[0]) for k, v in ks.items()}
    return out


def main():
    out, key, substr, alg = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4])
    e = {"kernels_matching": substr}
    for path in sys.argv[5:]:
        for c, ks in per_kernel_means(path, substr).items():
            e[c + "_KB_per_kernel"] = {k: round(v, 1) for k, v in ks.items()}
            e[c] = sum(ks.values())
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        e["traffic_bytes_corrected"] = (2 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024
        e["algorithmic_bytes"] = alg
        e["traffic_over_algorithmic"] = round(e["traffic_bytes_corrected"] / alg, 3)
    res = json.load(open(out)) if os.path.exists(out) else {}
    res[key] = e
    json.dump(res, open(out, "w"), indent=1)
    print(key, json.dumps(e))


if __name__ == "__main__":
    main()
