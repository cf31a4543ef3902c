"""Summarises rocprofv3 --pmc CSV output per kernel (average per dispatch)."""
import collections
import csv
import glob
import re
import sys


def kernel_key(name):
    m = re.search(r"(k_[a-z_]+)", name)
    return m.group(1) if m else name[:40]


def summarise(directory):
    """{kernel: {counter: average per dispatch}} of one pass directory."""
    out = {}
    for f in glob.glob(directory + "/*/*_counter_collection.csv") + glob.glob(directory + "/*_counter_collection.csv"):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        disp = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = kernel_key(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
        for k, v in agg.items():
            out.setdefault(k, {}).update({c: x / len(disp[k]) for c, x in v.items()})
            out[k]["dispatches"] = len(disp[k])
    return out


if __name__ == "__main__":
    for d in sys.argv[1:]:
        for k, v in sorted(summarise(d).items()):
            if any(w in k for w in ("march", "replay", "splat", "ingest", "classify", "hist")):
                print(d.rstrip("/").split("/")[-1], k, "x%d" % v.pop("dispatches"), {c: "%.4g" % x for c, x in sorted(v.items())})
