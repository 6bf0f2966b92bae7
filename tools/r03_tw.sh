#!/bin/bash
# transposed LDS twiddles of the last-axis kernels: full GPU suite, config 4 / headline bench, tiled-kernel variant A/B
set -e
O=gpurun_out/r03tw
mkdir -p $O
rm -f $O/shapes.txt
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1 || { tail -30 $O/gputest.log; exit 1; }
tail -2 $O/gputest.log
python3 bench.py --config 4 --no-cpu-baseline > $O/config4.json 2> $O/config4.err
python3 bench.py > $O/bench.json 2> $O/bench.err
echo bench done
export AB_NO_FFT=1
L=$PWD/libmultiviewnative_amd/lib
MVN_PRODUCT_SO=$L/libmvn_ab_twtiled.so timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fixed or config1 or config0 or deconvolve_vs_oracle" > $O/tests_tiled.log 2>&1 || { tail -30 $O/tests_tiled.log; exit 1; }
tail -1 $O/tests_tiled.log
for s in "256 256 256" "128 128 128" "64 64 64" "192 192 192" "64 256 256" "256 256 256"; do
  for v in libmultiviewnative.so libmvn_ab_twtiled.so; do
    echo "== $s $v" >> $O/shapes.txt
    MVN_PRODUCT_SO=$L/$v AB_SHAPE="$s" python3 tools/sweep.py "" >> $O/shapes.txt 2>&1
  done
done
grep -E "^==|view-iter" $O/shapes.txt | cut -c1-330
