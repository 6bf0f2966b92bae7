#!/bin/bash
# A/B of builds (MVN_PRODUCT_SO) over the non-power-of-two shapes
set -e
O=gpurun_out/r03rm
mkdir -p $O
rm -f $O/shapes.txt
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "walking or mixed_radix or config4_long or fixed" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
export AB_NO_FFT=1
L=$PWD/libmultiviewnative_amd/lib
for s in "64 1920 1920" "576 576 576" "640 640 640" "320 320 320" "288 288 288" "768 768 768" "96 960 960" "64 1920 1920" "576 576 576"; do
  for v in $(cd $L && ls libmultiviewnative.so libmvn_ab_*.so); do
    echo "== $s $v" >> $O/shapes.txt
    MVN_PRODUCT_SO=$L/$v AB_SHAPE="$s" python3 tools/sweep.py "" >> $O/shapes.txt 2>&1
  done
done
grep -E "^==|view-iter" $O/shapes.txt | cut -c1-330
AB_ARGS="--config 4" AB_STEPS=5 tools/ab_bench.sh - MVN_PRODUCT_SO=$L/libmvn_ab_rf0.so - MVN_PRODUCT_SO=$L/libmvn_ab_rf0.so
