#!/bin/bash
# Counters of the marches every tap of which matters: configuration c2 (512^3, 1920x1080) under the translucent
# table, reference-order march with skipping (what fast mode gives way to) and the plain one.
# usage: bash tools/dense_profile.sh <tag>   -> gpurun_out/dense_<tag>/
set -u
tag=${1:-q}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/dense_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for mode in 2 0; do
  args="--volume 512 --width 1920 --height 1080 --frames 6 --table translucent --mode $mode"
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt$mode" -o kt -- python3 "$root/tools/profile_frame.py" $args > "$out/kt$mode.log" 2>&1 || exit 1
  pass() {
    name=$1; shift
    timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$out/$name$mode" -o pmc -- python3 "$root/tools/profile_frame.py" $args > "$out/$name$mode.log" 2>&1 || exit 1
  }
  pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY
  pass sq2 SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_INSTS_BRANCH
  pass fetch FETCH_SIZE
  pass write WRITE_SIZE
  pass tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
  pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
  cd "$root"
  echo "mode $mode" >> "$out/summary.txt"
  python3 tools/kernel_times.py "$out/kt$mode" | grep march >> "$out/summary.txt"
  python3 tools/pmc_summary.py "$out/sq1$mode" "$out/sq2$mode" "$out/fetch$mode" "$out/write$mode" "$out/tcc$mode" "$out/tcp$mode" | grep march_exact >> "$out/summary.txt"
  cd /tmp
done
cat "$out/summary.txt"
