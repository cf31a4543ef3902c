#!/bin/bash
# Durations of the frame's kernels when nothing overlaps them (one frame in flight): kernel trace over
# tools/profile_frame.py.   usage: bash tools/alone_kernel_times.sh <tag>  -> gpurun_out/prof_<tag>/alone_kernel_stats.csv
set -u
tag=${1:-rX}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
export VR_TILE_ORDER=0   # tiles in image order, as in the benchmark's pipelined frames
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/alone" -o alone -- python3 "$root/tools/profile_frame.py" --frames 12 > "$out/alone.log" 2>&1 || { tail -5 "$out/alone.log"; exit 1; }
cp "$out"/alone/*/alone_kernel_stats.csv "$out/alone_kernel_stats.csv" 2>/dev/null || cp "$out"/alone/alone_kernel_stats.csv "$out/alone_kernel_stats.csv"
python3 "$root/tools/kernel_times.py" "$out/alone"
