#!/bin/bash
# Diagnostic: bench.py throughput under environment settings, one per argument ("VAR=value").
for kv in "$@"; do
  env "$kv" timeout -k 10 200 python -u bench.py --steps 120 --no-cpu-baseline 2>/tmp/bench_env.err | tail -1 > /tmp/bench_env.json || { tail -5 /tmp/bench_env.err; exit 1; }
  python3 - "$kv" <<'PY'
import json, sys
d = json.load(open("/tmp/bench_env.json"))
print(sys.argv[1], "Gsamples/s %.0f" % d["value"], "ms/frame %.3f" % d["ms_per_step"], "fast march ms %.3f" % d["roofline"].get("kernel_ms_avg", 0),
      "single %.3f" % d["single_frame_ms"]["mean"])
PY
done
