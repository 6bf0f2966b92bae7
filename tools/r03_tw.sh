#!/bin/bash
# after the last-axis / split-window kernel changes: full GPU suite, config 4 / 3 / 1 and headline bench, shapes
set -e
O=gpurun_out/r03tw
mkdir -p $O
rm -f $O/shapes.txt
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1 || { tail -30 $O/gputest.log; exit 1; }
tail -2 $O/gputest.log
python3 bench.py --config 4 --no-cpu-baseline > $O/config4.json 2> $O/config4.err
python3 bench.py --config 3 --no-cpu-baseline > $O/config3.json 2> $O/config3.err
python3 bench.py --config 1 --no-cpu-baseline > $O/config1.json 2> $O/config1.err
python3 bench.py > $O/bench.json 2> $O/bench.err
echo bench done
export AB_NO_FFT=1
for s in "576 576 576" "640 640 640" "768 768 768" "1024 1024 1024" "320 1920 1920"; do
  echo "== $s" >> $O/shapes.txt
  AB_SHAPE="$s" python3 tools/sweep.py "" >> $O/shapes.txt 2>&1
done
grep -E "^==|view-iter" $O/shapes.txt | cut -c1-330
