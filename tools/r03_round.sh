#!/bin/bash
# Round-3 GPU session: full GPU suite, the bench line, kernel-trace statistics, the sharded step's
# timeline on one rank.  Outputs under gpurun_out/r03/.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r03
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/gputest.log 2>&1 || { tail -30 $OUT/gputest.log; exit 1; }
tail -3 $OUT/gputest.log
timeout -k 10 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
echo "bench done"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $OUT/stats -o p -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-side > $OUT/stats_bench.json 2> $OUT/stats.err || { tail -20 $OUT/stats.err; exit 1; }
echo "stats pass done"
timeout -k 10 600 rocprofv3 --kernel-trace -d $OUT/tl -o p -- python3 $ROOT/bench.py --gpus 1 --force-dist --backend nccl --chunks 4 --steps 6 --warmup 2 --no-side > $OUT/tl_bench.json 2> $OUT/tl.err || { tail -20 $OUT/tl.err; exit 1; }
echo "timeline pass done"
cd $ROOT
python3 tools/rocpd_stats.py stats $OUT/stats $OUT/kernel_stats.csv > /dev/null
python3 tools/timeline.py $OUT/tl 2 > $OUT/overlap_timeline.md || true
timeout -k 10 300 python bench.py --simultaneous --no-cpu-baseline > $OUT/simultaneous_1rank_bench.json 2>> $OUT/bench.err
echo "all done"
