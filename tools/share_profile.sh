#!/bin/bash
# Kernel times of one sort-first part's share of the benchmark frame.  usage: bash tools/share_profile.sh <parts>
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/share_$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 $root/tools/profile_frame.py --frames 8 --parts $1 > $out/kt.log 2>&1 || exit 1
python3 $root/tools/kernel_times.py $out/kt | grep -v "phantom\|build_bricks\|swap_max\|classify"
