#!/bin/bash
# Kernel trace of the pipelined benchmark run and the start/end of the frame's kernels over a few frames
# in the middle of the timed region.   usage: bash tools/pipeline_trace.sh [bench.py options]
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pipeline_trace
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$out" -o trace -- python3 "$root/bench.py" --no-cpu-baseline --steps 120 "$@" > "$out/bench.json" 2> "$out/bench.err" || { tail -5 "$out/bench.err"; exit 1; }
python3 - "$out" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
ev = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    k = None
    if "k_march_fast<" in n and ", false, " in n: k = "march"
    elif "k_march_long<" in n and ", false, " in n: k = "long"
    elif "k_replay_exact<" in n and ", false, " in n: k = "replay"
    elif "k_splat" in n: k = "splat"
    elif "k_order_tiles" in n: k = "order"
    elif "fillBuffer" in n: k = "fill"
    if k: ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k, r["Queue_Id"]))
ev.sort()
mid = len(ev) // 2
while ev[mid][2] != "march": mid += 1
t0 = ev[mid][0]
for s, e, k, q in ev[mid:mid + 40]:
    print("%9.1f %9.1f  %7.1f us  %-7s queue %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, k, q))
PY
