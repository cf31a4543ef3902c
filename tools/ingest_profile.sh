#!/bin/bash
# Kernel time and HBM traffic of the volume upload's kernels on the GPU box.
# usage: bash tools/ingest_profile.sh <tag> [edge]   -> gpurun_out/ingest_<tag>/
set -u
tag=${1:-q}
edge=${2:-2048}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/ingest_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -o kt -- python3 "$root/tools/ingest_time.py" "$edge" > "$out/kt.log" 2>&1 || exit 1
pass() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -o pmc -- python3 "$root/tools/ingest_time.py" "$edge" > "$out/$name.log" 2>&1 || exit 1
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY
pass sq2 SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT
cd "$root"
python3 tools/kernel_times.py "$out/kt" | tee "$out/kernel_times.txt"
python3 tools/pmc_summary.py "$out/tcc" "$out/fetch" "$out/write" "$out/sq1" "$out/sq2" | tee "$out/pmc.txt"
