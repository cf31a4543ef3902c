#!/bin/bash
# Diagnostic: kernel time and fetched bytes of the upload's re-tiling kernel for several builds
# (make variant NAME=x DEFS=...) and walked layers.  usage: bash tools/ingest_sweep.sh "<lib variants>" "<layers>" [edge]
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
edge=${3:-2048}
out=$root/gpurun_out/ingest_sweep
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for v in $1; do
  lib=libvr_hip_$v.so
  [ "$v" = default ] && lib=libvr_hip.so
  for b in $2; do
    export VR_LIB_NAME=$lib VR_TEST_INGEST_LAYERS=$b
    rm -rf "$out/kt_${v}_$b" "$out/fe_${v}_$b"
    timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt_${v}_$b" -o kt -- python3 "$root/tools/ingest_time.py" "$edge" > "$out/kt_${v}_$b.log" 2>&1 || { tail -5 "$out/kt_${v}_$b.log"; exit 1; }
    timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/fe_${v}_$b" -o pmc -- python3 "$root/tools/ingest_time.py" "$edge" > "$out/fe_${v}_$b.log" 2>&1 || { tail -5 "$out/fe_${v}_$b.log"; exit 1; }
    echo "variant $v layers $b: $(python3 "$root/tools/kernel_times.py" "$out/kt_${v}_$b" | grep gest_bricks | sed "s/.*calls/calls/") $(python3 "$root/tools/pmc_summary.py" "$out/fe_${v}_$b" | grep ingest | sed 's/.*{/{/')"
  done
done
