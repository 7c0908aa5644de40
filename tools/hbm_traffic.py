"""Writes the hbm_traffic JSON that bench.py reads (profiles/rN_vM_hbm_traffic[_configC].json) from the two counter
passes of tools/profile_bench.sh: the FIRST dispatch of the render kernel in each pass is the timed launch (the second
is the counting launch that bench.py makes outside the timed region).
    python tools/hbm_traffic.py gpurun_out/prof_<tag> <config>"""
import csv
import glob
import json
import sys

out, cfg = sys.argv[1], int(sys.argv[2])


def first_dispatch(d, counter):
    per = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if ("render_kernel" in r["Kernel_Name"] or "render_samples_kernel" in r["Kernel_Name"]) and r["Counter_Name"] == counter:
                k = int(r["Dispatch_Id"])
                per.setdefault(k, [r["Kernel_Name"], 0.0])
                per[k][1] += float(r["Counter_Value"])
    k = min(per)
    return per[k]


bench = json.loads(open(out + "/bench.json").read().strip().splitlines()[-1])
name, fetch_kb = first_dispatch(out + "/fetch", "FETCH_SIZE")
_, write_kb = first_dispatch(out + "/write", "WRITE_SIZE")
import re
m = re.search(r"(\d+) spp (?:in total|per GPU)", bench["config"]["workload"])
d = re.search(r"(\d+) bounces", bench["config"]["workload"])
# FETCH_SIZE tallies every memory-side read request at 64 bytes. Wide coalesced reads (the film) go out as 128-byte
# requests, so their bytes are twice the reported figure; the 16-byte gathers of a walk through a big mesh go out as 64-byte
# requests and are reported exactly (profiles/r3_fetch_calibration.json: tools/ubench/fetch_calib.hip, known byte counts).
factor = 1.0 if cfg == 3 else 2.0
print(json.dumps({
    "kernel": name.split("(")[0], "config": cfg, "workload": bench["config"]["workload"],
    "spp": int(m.group(1)), "depth": int(d.group(1)),
    "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb, "fetch_factor": factor,
    "bytes_per_launch": factor * fetch_kb * 1024 + write_kb * 1024,
    "bytes_per_launch_fetch_as_reported": fetch_kb * 1024 + write_kb * 1024,
    "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate runs of `python3 bench.py --config %d --no-cpu --steps 1 "
           "--warmup 0` (tools/profile_bench.sh), first render kernel dispatch of each; bytes = fetch_factor * FETCH_SIZE_KB * 1024 + "
           "WRITE_SIZE_KB * 1024. fetch_factor: 2 where the film's coalesced reads dominate (128-byte requests tallied at 64), 1 for "
           "configuration 3, whose reads are 16-byte gathers that leave L2 as 64-byte requests (profiles/r3_fetch_calibration.json: "
           "64-byte records read by four 16-byte loads per lane report 0.95-1.00 of their bytes)" % cfg}, indent=1))
