#!/bin/bash
# A/B two builds over several shapes on the same box: $1 = alternative .so under lib/
set -e
cd "$(dirname "$0")/.."
L=libmultiviewnative_amd/lib
cp $L/libmultiviewnative.so /tmp/_A.so
for shp in "576 576 576" "640 640 640" "1024 1024 1024" "320 1920 1920"; do
  echo "== A $shp"; python tools/shape_probe.py $shp 31 | grep view-iter
  cp $L/$1 $L/libmultiviewnative.so
  echo "== B $shp"; python tools/shape_probe.py $shp 31 | grep view-iter
  cp /tmp/_A.so $L/libmultiviewnative.so
done
