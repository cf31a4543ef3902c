#!/bin/bash
# Collects the HBM byte counters of the lit workload (two passes, one counter each).
# usage: bash tools/collect_shaded_traffic.sh <tag>   -> gpurun_out/prof_<tag>_shaded/...
set -u
tag=${1:-rX}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_${tag}_shaded
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
export VR_TILE_ORDER=0   # tiles in image order, as in the benchmark's pipelined frames
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -o pmc -- python3 "$root/tools/profile_frame.py" --frames 6 --shading > "$out/fetch.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/write" -o pmc -- python3 "$root/tools/profile_frame.py" --frames 6 --shading > "$out/write.log" 2>&1 || exit 1
cd "$root"
python3 tools/pmc_summary.py "$out/fetch" "$out/write" > "$out/pmc_summary.txt"
cat "$out/pmc_summary.txt"
