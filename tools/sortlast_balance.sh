#!/bin/bash
# Per-box work of the sort-last split with even boxes and with boxes cut where the occupied bricks balance
# (one GPU stands in for all eight, one box after the other).
# usage: bash tools/sortlast_balance.sh <volume edge> [out dir]  -> <out dir>/sortlast_rank_<edge>_{even,occupied}.json
set -u
edge=${1:-2048}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=${2:-$root/gpurun_out}
mkdir -p "$out"
cd "$root"
for b in even occupied; do
  f=$out/sortlast_rank_${edge}_$b.json
  timeout -k 10 500 python3 tools/sortlast_rank_time.py --volume "$edge" --frames 8 --step-deg 45 --balance $b > "$f" 2> "$out/sortlast_rank_${edge}_$b.err" || { tail -5 "$out/sortlast_rank_${edge}_$b.err"; exit 1; }
  python3 - "$f" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(j["balance"], [round(b["partial_ms_mean"], 2) for b in j["boxes"]], "max/mean %.2f" % j["partial_ms_max_over_mean"],
      "max of max %.2f" % max(b["partial_ms_max"] for b in j["boxes"]))
print("   ", [b["bricks"] for b in j["boxes"]])
PY
done
