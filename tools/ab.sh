#!/bin/bash
# A/B two builds of the library on the same box: $1 = alternative .so (relative to lib/)
set -e
cd "$(dirname "$0")/.."
L=libmultiviewnative_amd/lib
echo "== A: default build"; python tools/sweep.py ""
cp $L/libmultiviewnative.so /tmp/_A.so; cp $L/$1 $L/libmultiviewnative.so
echo "== B: $1"; python tools/sweep.py ""
cp /tmp/_A.so $L/libmultiviewnative.so
echo "== A again, lambda = 0"; SWEEP_LAMBDA=0 python tools/sweep.py ""
