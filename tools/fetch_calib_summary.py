"""Divides the bytes each launch of tools/ubench/fetch_calib requested by what rocprofv3's FETCH_SIZE reported for it, and
prints one JSON object (committed as profiles/rN_fetch_calibration.json).
    python tools/fetch_calib_summary.py <stdout of the program> <dir of the FETCH_SIZE pass> <dir of the TCC_HIT / TCC_MISS pass>"""
import csv
import glob
import json
import sys


def per_dispatch(d, counters):
    rows = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "calib_" in r["Kernel_Name"] and r["Counter_Name"] in counters:
                k = int(r["Dispatch_Id"])
                rows.setdefault(k, {"kernel": r["Kernel_Name"].split("(")[0]})
                rows[k][r["Counter_Name"]] = rows[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return [rows[k] for k in sorted(rows)]


launches = [l.split() for l in open(sys.argv[1]) if l.startswith("table")]
fetch = per_dispatch(sys.argv[2], {"FETCH_SIZE"})
hits = per_dispatch(sys.argv[3], {"TCC_HIT_sum", "TCC_MISS_sum"}) if len(sys.argv) > 3 else [None] * len(fetch)
out = []
for l, f, h in zip(launches, fetch, hits):
    req = float(l[5])
    row = {"table_MiB": int(l[1]), "pattern": l[3], "kernel": f["kernel"], "requested_bytes": req, "FETCH_SIZE_KB": f["FETCH_SIZE"],
           "requested_over_reported": req / (f["FETCH_SIZE"] * 1024.0)}
    if h:
        miss = h["TCC_MISS_sum"] / max(1.0, h["TCC_MISS_sum"] + h["TCC_HIT_sum"])
        row["l2_miss_fraction"] = miss
        row["missed_request_bytes_over_reported"] = req * miss / (f["FETCH_SIZE"] * 1024.0)
    out.append(row)
print(json.dumps({"what": "rocprofv3 FETCH_SIZE against known byte counts on gfx950 (tools/ubench/fetch_calib.hip, one launch per row)", "rows": out}, indent=1))
