#!/bin/bash
# A/B of alternative builds on the same box: args = .so files under lib/ (optionally VAR=value
# assignments first, applied to every run); prints per-kernel times of one view update at 512^3
# (tools/sweep.py) for the default build and every variant
cd "$(dirname "$0")/.."
L=libmultiviewnative_amd/lib
cp $L/libmultiviewnative.so /tmp/_A.so
run() { python tools/sweep.py "" 2>&1 | grep view-iter; }
echo "== A: default build"; run
echo "== A with MVN_NO_WAVE_ROWS=1"; MVN_NO_WAVE_ROWS=1 run
for v in "$@"; do
  cp $L/$v $L/libmultiviewnative.so
  echo "== variant: $v"; run
  echo "== variant: $v with MVN_NO_WAVE_ROWS=1"; MVN_NO_WAVE_ROWS=1 run
  cp /tmp/_A.so $L/libmultiviewnative.so
done
echo "== A again"; run
