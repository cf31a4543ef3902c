#!/bin/bash
# Diagnostic: bench.py throughput for several builds of the library (make variant NAME=x ...).
# usage: bash tools/variant_bench.sh default x y ...
for v in "$@"; do
  lib=libvr_hip_$v.so
  [ "$v" = default ] && lib=libvr_hip.so
  VR_LIB_NAME=$lib timeout -k 10 200 python -u bench.py --steps 120 --no-cpu-baseline 2>/tmp/bench_$v.err | tail -1 > /tmp/bench_$v.json || { tail -5 /tmp/bench_$v.err; exit 1; }
  python3 - "$v" <<'PY'
import json, sys
v = sys.argv[1]
d = json.load(open("/tmp/bench_%s.json" % v))
print("build", v, "Gsamples/s %.0f" % d["value"], "ms/frame %.3f" % d["ms_per_step"], "fast march ms %.3f" % d["roofline"].get("kernel_ms_avg", 0),
      "single %.3f" % d["single_frame_ms"]["mean"])
PY
done
