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
            if "render_kernel" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                k = int(r["Dispatch_Id"])
                per.setdefault(k, [r["Kernel_Name"], 0.0])
                per[k][1] += float(r["Counter_Value"])
    k = min(per)
    return per[k]


bench = json.loads(open(out + "/bench.json").read().strip().splitlines()[-1])
name, fetch_kb = first_dispatch(out + "/fetch", "FETCH_SIZE")
_, write_kb = first_dispatch(out + "/write", "WRITE_SIZE")
import re
m = re.search(r"(\d+) spp", bench["config"]["workload"])
d = re.search(r"(\d+) bounces", bench["config"]["workload"])
print(json.dumps({
    "kernel": name.split("(")[0], "config": cfg, "workload": bench["config"]["workload"],
    "spp": int(m.group(1)), "depth": int(d.group(1)),
    "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
    "bytes_per_launch": 2 * fetch_kb * 1024 + write_kb * 1024,
    "bytes_per_launch_fetch_as_reported": fetch_kb * 1024 + write_kb * 1024,
    "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate runs of `python3 bench.py --config %d --no-cpu --steps 1 "
           "--warmup 0` (tools/profile_bench.sh), first render_kernel dispatch of each; bytes = 2 * FETCH_SIZE_KB * 1024 (gfx950 reports "
           "half the bytes of wide coalesced reads; for the 16-byte gathers of a big mesh that factor is uncalibrated, so the as-reported "
           "sum is given too) + WRITE_SIZE_KB * 1024" % cfg}, indent=1))
