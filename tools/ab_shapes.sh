#!/bin/bash
# A/B two builds over several shapes on the same box: $1 = alternative .so under lib/, rest = shapes "d0 d1 d2"
set -e
cd "$(dirname "$0")/.."
L=libmultiviewnative_amd/lib
V=$1; shift
cp $L/libmultiviewnative.so /tmp/_A.so
for shp in "$@"; do
  echo "== A $shp"; python tools/shape_probe.py $shp 31 | grep view-iter
  cp $L/$V $L/libmultiviewnative.so
  echo "== B $shp"; python tools/shape_probe.py $shp 31 | grep view-iter
  cp /tmp/_A.so $L/libmultiviewnative.so
done
