#!/bin/bash
# Diagnostic: bench.py throughput for several values of the per-lane work budget (VR_LANE_BUDGET).
# usage: [CONFIG=c4] bash tools/budget_sweep.sh [lib name] 64 80 96 ...
lib=libvr_hip.so
case "$1" in *.so) lib=$1; shift;; esac
cfg=${CONFIG:-c4}
for b in "$@"; do
  VR_LIB_NAME=$lib VR_LANE_BUDGET=$b timeout -k 10 200 python -u bench.py --config $cfg --steps 120 --no-cpu-baseline 2>/tmp/bench_b$b.err | tail -1 > /tmp/bench_b$b.json || { tail -5 /tmp/bench_b$b.err; exit 1; }
  python3 - "$b" "$lib" "$cfg" <<'PY'
import json, sys
b = sys.argv[1]
d = json.load(open("/tmp/bench_b%s.json" % b))
print(sys.argv[3], sys.argv[2], "budget", b, "ms/frame %.3f" % d["ms_per_step"], "fast march ms %.3f" % d["roofline"].get("kernel_ms_avg", 0),
      "single %.3f" % d["single_frame_ms"]["mean"], "handed over %.3f" % d["roofline"]["handed_over_ray_fraction"])
PY
done
