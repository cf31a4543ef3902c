#!/bin/bash
# Diagnostic: bench.py throughput for several lane budgets.
for b in "$@"; do
  VR_LANE_BUDGET=$b timeout -k 10 200 python -u bench.py --steps 60 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/bench_$b.json || exit 1
  python3 - "$b" <<'PY'
import json, sys
b = sys.argv[1]
d = json.load(open("/tmp/bench_%s.json" % b))
print("budget", b, "Gsamples/s %.0f" % d["value"], "ms/frame %.3f" % d["ms_per_step"], "fast march ms %.3f" % d["roofline"].get("kernel_ms_avg", 0))
PY
done
