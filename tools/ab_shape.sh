#!/bin/bash
# A/B of alternative builds on one shape: $1 $2 $3 = d0 d1 d2, rest = .so files under lib/
cd "$(dirname "$0")/.."
L=libmultiviewnative_amd/lib
S="$1 $2 $3"; shift 3
cp $L/libmultiviewnative.so /tmp/_A.so
run() { python tools/shape_probe.py $S 31 2>&1 | grep view-iter; }
echo "== A: default build"; run
for v in "$@"; do
  cp $L/$v $L/libmultiviewnative.so
  echo "== variant: $v"; run
  cp /tmp/_A.so $L/libmultiviewnative.so
done
echo "== A again"; run
