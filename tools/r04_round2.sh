#!/bin/bash
# Round-4 GPU session, part 2: wall time of the default bench command, cost of one slab of an N-device run, power
# and clock beside the loop (default build, Nyquist launches of their own, FFT dim0 leg).
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r04
mkdir -p $OUT
cd $ROOT
( time timeout -k 10 500 python bench.py > $OUT/bench8.json 2> $OUT/bench8.err ) 2> $OUT/bench8.time; cat $OUT/bench8.time | tail -3
timeout -k 10 400 python tools/slab_cost.py > $OUT/slab_cost.txt 2> $OUT/slab_cost.err; cat $OUT/slab_cost.txt
for cfg in "" "MVN_NYQ_RIDE=0" "MVN_DIM0_DIRECT=0"; do echo "== ${cfg:-default}"; timeout -k 10 200 tools/power_probe.sh $cfg; done > $OUT/power_probe.txt 2>&1; cat $OUT/power_probe.txt | cut -c1-700
echo "all done"
