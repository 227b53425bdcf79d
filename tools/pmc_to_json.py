#!/usr/bin/env python3
"""gpurun_out/pmc_<tag>/pass*/**/*counter_collection.csv  ->  profiles/<out>.json
Per kernel: mean FETCH_SIZE / WRITE_SIZE (KB, as rocprofv3 reports them) per launch and the derived HBM byte counts.
usage: pmc_to_json.py <tag> <frames_per_step> <out.json> [calib_tag]"""
import csv
import glob
import json
import sys
from collections import defaultdict


def load(tag):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob("gpurun_out/pmc_%s/pass*/**/*counter_collection.csv" % tag, recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("av1mi::", "")
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


tag, frames, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes (tools/prof_pmc.sh %s)" % tag,
       "frames_per_step": frames, "kernels": {}, "calibration": None}
read_corr = {16: 2.0, 8: None}
if len(sys.argv) > 4:
    cal = load(sys.argv[4])
    n = 128 << 20
    q, dq = cal.get("k_quantize", {}), cal.get("k_dequantize", {})
    res["calibration"] = {
        "k_quantize":   {"known_read_bytes": n * 4, "FETCH_SIZE_bytes": q.get("FETCH_SIZE", 0) * 1024, "known_write_bytes": n * 2, "WRITE_SIZE_bytes": q.get("WRITE_SIZE", 0) * 1024},
        "k_dequantize": {"known_read_bytes": n * 2, "FETCH_SIZE_bytes": dq.get("FETCH_SIZE", 0) * 1024, "known_write_bytes": n * 4, "WRITE_SIZE_bytes": dq.get("WRITE_SIZE", 0) * 1024}}
    if q.get("FETCH_SIZE"):
        read_corr[16] = n * 4 / (q["FETCH_SIZE"] * 1024)
    if dq.get("FETCH_SIZE"):
        read_corr[8] = n * 2 / (dq["FETCH_SIZE"] * 1024)
    res["calibration"]["read_correction_16B_per_lane"] = read_corr[16]
    res["calibration"]["read_correction_8B_per_lane"] = read_corr[8]
for k, d in load(tag).items():
    if "FETCH_SIZE" not in d and "WRITE_SIZE" not in d:
        continue
    res["kernels"][k] = {"FETCH_SIZE_KB_per_launch": d.get("FETCH_SIZE"), "WRITE_SIZE_KB_per_launch": d.get("WRITE_SIZE"),
                         "fetch_bytes_uncorrected": d.get("FETCH_SIZE", 0) * 1024, "write_bytes": d.get("WRITE_SIZE", 0) * 1024,
                         "sq_counters_per_launch": {c: v for c, v in sorted(d.items()) if c not in ("FETCH_SIZE", "WRITE_SIZE")}}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
