#!/bin/bash
# Variant of the library that differs from the current build ONLY in the flags of the two split-operand edge kernels
# (objects of every other file are taken from the Makefile build): tools/edge_variant.sh <name> "<flags>"
set -e
NAME=$1; FL=$2
cd "$(dirname "$0")/../nmrgnn_amd/csrc"
OUT=/tmp/ngev_$NAME; mkdir -p $OUT variants
for s in edge_bwd_h2 edge_fwd_h2; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize -I../../include $FL -c $s.hip -o $OUT/$s.o &
done
wait
OBJS=$(ls *.o | grep -v "edge_bwd_h2.o\|edge_fwd_h2.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $OUT/edge_bwd_h2.o $OUT/edge_fwd_h2.o -ldl -o variants/$NAME.so
ls -la variants/$NAME.so
