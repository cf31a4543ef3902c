"""Average duration of the lane march over the TIMED region of a default `bench.py` run, out of the rocprofv3
kernel trace of that run.  The stats file of the same run averages every dispatch of the program: the
set-up's tuning runs, the timed frames, and the pass that times single frames afterwards (which launches
its tiles longest first and is faster); the HIP events inside bench.py cover the timed frames only.
The program's order is: tuning runs, warm-up, `steps` timed frames, `steps` single frames.

  python tools/timed_region_avg.py <rocprofv3 output dir> [steps]
"""
import csv
import glob
import sys

steps = int(sys.argv[2]) if len(sys.argv) > 2 else 360
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_march_fast<" in r["Kernel_Name"] and ", false, " in r["Kernel_Name"].split("(")[0]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
assert len(dur) >= 2 * steps, "not a default run: %d dispatches" % len(dur)
timed, single = dur[-2 * steps:-steps], dur[-steps:]
print("dispatches of the lane march: %d (all: avg %.4f ms)" % (len(dur), sum(dur) / len(dur)))
print("timed region, %d pipelined frames: avg %.4f ms  min %.4f  max %.4f" % (steps, sum(timed) / steps, min(timed), max(timed)))
print("single frames afterwards, %d: avg %.4f ms" % (steps, sum(single) / steps))
