"""Summarise the rocprofv3 output of tools/profile_gpu.sh for one kernel name substring:
per-launch averages of every counter, plus corrected HBM traffic (FETCH_SIZE on gfx950 counts
half the bytes of a wide coalesced stream: MI355X_MICROARCH.md, HBM section -> doubled here).
usage: python tools/pmc_summary.py gpurun_out/prof_<tag> k_check_edges [out.json]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    root, kern = sys.argv[1], sys.argv[2]
    res = {}
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(list)
        with open(path) as f:
            for row in csv.DictReader(f):
                if kern in row["Kernel_Name"] and (kern + "_f64" not in row["Kernel_Name"] or kern.endswith("_f64")):
                    acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, v in acc.items():
            res[k] = sum(v) / len(v)
    for path in glob.glob(os.path.join(root, "**", "*kernel_stats.csv"), recursive=True):
        if "trace_streams" in path:  # (the bench line's own mode, kernels of several streams overlapping: not a kernel alone)
            continue
        total_ns, calls = 0.0, 0
        with open(path) as f:
            for row in csv.DictReader(f):  # (a template's instantiations are rows of their own: one kernel to this summary)
                if kern in row["Name"] and (kern + "_f64" not in row["Name"] or kern.endswith("_f64")):
                    total_ns += float(row["TotalDurationNs"])
                    calls += int(row["Calls"])
        if calls:
            res["avg_ns"] = total_ns / calls
            res["calls"] = calls
    if "FETCH_SIZE" in res or "WRITE_SIZE" in res:
        # rocprofv3 reports KiB; FETCH_SIZE x2 per the gfx950 correction
        res["hbm_bytes_per_launch"] = (2 * res.get("FETCH_SIZE", 0.0) + res.get("WRITE_SIZE", 0.0)) * 1024
    print(json.dumps(res, indent=1))
    if len(sys.argv) > 3:
        with open(sys.argv[3], "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
