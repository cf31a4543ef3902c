#!/bin/bash
# Diagnostic: upload time under several builds of the library (make variant NAME=x DEFS=...).
# usage: bash tools/ingest_variants.sh default x y ...
for v in "$@"; do
  lib=libvr_hip_$v.so
  [ "$v" = default ] && lib=libvr_hip.so
  echo "build $v: $(VR_LIB_NAME=$lib INGEST_REPS=8 timeout -k 10 100 python3 tools/ingest_time.py 2048 2>/dev/null | grep edge)"
done
