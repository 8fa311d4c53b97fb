#!/bin/bash
# Builds nmrgnn_amd/csrc/variants/<name>.so from the CURRENT sources with extra compiler flags, for same-box A/B runs
# (tools/ab.sh).  Usage: tools/build_variant.sh <name> ["extra flags for every file"] ["file.hip=flags" ...]
set -e
NAME=$1; EXTRA=$2; shift; shift || true
cd "$(dirname "$0")/../nmrgnn_amd/csrc"
SRCS=$(ls *.hip)
OUT=/tmp/ngvar_$NAME; mkdir -p $OUT variants
pids=()
for s in $SRCS; do
  fl="$EXTRA"
  for kv in "$@"; do [ "${kv%%=*}" = "$s" ] && fl="$fl ${kv#*=}"; done
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize -I../../include $fl -c $s -o $OUT/${s%.hip}.o &
  pids+=($!)
  if [ ${#pids[@]} -ge 8 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OUT/*.o -o variants/$NAME.so
ls -la variants/$NAME.so
