#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box: kernel stats of bench.py and the PMC passes
# over tools/profile_frame.py (one pass per counter group, each a run of its own).
# usage: bash tools/collect_profiles.sh <tag>   -> gpurun_out/prof_<tag>/...
set -u
tag=${1:-rX}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/bench" -o bench -- python3 "$root/bench.py" --no-cpu-baseline > "$out/bench_under_rocprof.json" 2> "$out/bench.err" || exit 1
# the benchmark's pipelined frames launch their tiles in image order; profile_frame.py renders one frame at a
# time, which would launch them longest first: switched off so that the counters describe the benchmark's launches
export VR_TILE_ORDER=0
pass() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -o pmc -- python3 "$root/tools/profile_frame.py" --frames 6 > "$out/$name.log" 2>&1 || exit 1
}
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY
pass sq2 SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_INSTS_BRANCH
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
cd "$root"
python3 tools/pmc_summary.py "$out/sq1" "$out/sq2" "$out/fetch" "$out/write" "$out/tcc" > "$out/pmc_summary.txt"
# afterwards, in the repository: python tools/make_traffic_json.py gpurun_out/prof_$tag 2048_3840x2160_fast profiles/$tag
cp "$out"/bench/*/bench_kernel_stats.csv "$out/bench_kernel_stats.csv" 2>/dev/null || cp "$out"/bench/bench_kernel_stats.csv "$out/bench_kernel_stats.csv"
tail -1 "$out/bench_under_rocprof.json"
cat "$out/pmc_summary.txt"
