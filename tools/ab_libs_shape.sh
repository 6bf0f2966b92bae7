#!/bin/bash
# Same-box A/B of variant builds on ONE shape (one view update, per-kernel events; tools/shape_probe.py):
#   tools/ab_libs_shape.sh "64 1920 1920 31" name1 name2 ...      ("-" = the product library itself)
cd "$(dirname "$0")/.."
SHAPE=$1; shift
for v in - "$@" - "$@"; do
  so=libmultiviewnative_amd/lib/libmultiviewnative.so
  [ "$v" != "-" ] && so=libmultiviewnative_amd/lib/libmultiviewnative_$v.so
  echo -n "== $v  "
  MVN_PRODUCT_SO=$PWD/$so timeout -k 10 300 python tools/shape_probe.py $SHAPE 2>/dev/null | grep view-iter | cut -c1-330
done
