"""Sums rocprofv3 --pmc counter_collection.csv files per dispatch of the render kernel.
Usage: python tools/pmc_summary.py <dir> [<dir> ...]"""
import collections
import csv
import glob
import sys

for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            if "render_kernel" not in r["Kernel_Name"] and "render_samples_kernel" not in r["Kernel_Name"]:
                continue
            k = (r["Dispatch_Id"], r["Grid_Size"], r["VGPR_Count"], r["LDS_Block_Size"])
            agg.setdefault(k, collections.OrderedDict())
            agg[k][r["Counter_Name"]] = agg[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        for k, c in agg.items():
            print(f, "dispatch", k[0], "grid", k[1], "vgpr", k[2], "lds", k[3])
            for n, v in c.items():
                print("    %-28s %.6g" % (n, v))
