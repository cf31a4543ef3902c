"""Summarises rocprofv3 --pmc CSV output per kernel (average per dispatch)."""
import collections
import csv
import glob
import sys

for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*_counter_collection.csv") + glob.glob(d + "/*_counter_collection.csv"):
        rows = list(csv.DictReader(open(f)))
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        disp = collections.defaultdict(set)
        for r in rows:
            k = r["Kernel_Name"].split("(")[0][-40:]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
        for k, v in agg.items():
            if "march" in k or "replay" in k or "splat" in k:
                n = len(disp[k])
                print(d, k, "x%d" % n, {c: "%.4g" % (x / n) for c, x in sorted(v.items())})
