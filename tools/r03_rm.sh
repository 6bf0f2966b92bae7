#!/bin/bash
# A/B of builds (MVN_PRODUCT_SO) over shapes with long dim1 lines
set -e
O=gpurun_out/r03rm
mkdir -p $O
rm -f $O/shapes.txt
L=$PWD/libmultiviewnative_amd/lib
for v in $(cd $L && ls libmvn_ab_*.so); do
MVN_PRODUCT_SO=$L/$v timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "long_lines or mixed_radix or config4_long" > $O/tests_$v.log 2>&1 || { tail -30 $O/tests_$v.log; exit 1; }
tail -1 $O/tests_$v.log
done
export AB_NO_FFT=1
for s in "64 1920 1920" "32 1280 1280" "64 1920 1920" "32 1280 1280"; do
  for v in $(cd $L && ls libmultiviewnative.so libmvn_ab_*.so); do
    echo "== $s $v" >> $O/shapes.txt
    MVN_PRODUCT_SO=$L/$v AB_SHAPE="$s" python3 tools/sweep.py "" >> $O/shapes.txt 2>&1
  done
done
grep -E "^==|view-iter" $O/shapes.txt | cut -c1-330
