#!/bin/bash
# A/B of alternative builds on the same box: args = .so files under lib/ (built next to the default
# one with other -D flags); prints per-kernel times of one view update at 512^3 (tools/sweep.py) for
# the default build and every variant, default first and last.  AB_SHAPE="d0 d1 d2" for other volumes.
cd "$(dirname "$0")/.."
L=libmultiviewnative_amd/lib
cp $L/libmultiviewnative.so /tmp/_A.so
run() { python tools/sweep.py "" 2>&1 | grep view-iter; }
echo "== A: default build"; run
for v in "$@"; do
  cp $L/$v $L/libmultiviewnative.so
  echo "== variant: $v"; run
  cp /tmp/_A.so $L/libmultiviewnative.so
done
echo "== A again"; run
