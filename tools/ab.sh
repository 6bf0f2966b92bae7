#!/bin/bash
# A/B builds of the library on the same box: args = alternative .so files under lib/
set -e
cd "$(dirname "$0")/.."
L=libmultiviewnative_amd/lib
cp $L/libmultiviewnative.so /tmp/_A.so
echo "== A: default build"; python tools/sweep.py ""
for v in "$@"; do
  cp $L/$v $L/libmultiviewnative.so
  echo "== variant: $v"; python tools/sweep.py ""
  cp /tmp/_A.so $L/libmultiviewnative.so
done
