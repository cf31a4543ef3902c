"""Diagnostic: per-kernel averages from a rocprofv3 --kernel-trace --stats output directory."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"].split("(")[0][-44:]
    if float(r["Percentage"]) > 0.3:
        print("%-44s calls %5s  avg %8.4f ms  min %7.3f  max %7.3f  %5s%%" % (
            n, r["Calls"], float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6, r["Percentage"]))
