#!/bin/bash
# Round-4 GPU session, part 1: kernel-trace statistics of the bench command, SQ and HBM-traffic counter passes
# (tools/profile_round.sh), the other BASELINE configs, the halo mode through the Python driver on one rank (plain
# and through RCCL send / recv with itself).  Outputs under gpurun_out/r04/.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r04
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 tools/profile_round.sh r04 > $OUT/profile_round.log 2>&1 || { tail -5 $OUT/profile_round.log; echo "profile_round failed"; exit 1; }
echo "profile_round done"
for c in 1 3 4; do
  timeout -k 10 400 python bench.py --config $c --no-cpu-baseline --no-abi > $OUT/config${c}_bench.json 2> $OUT/config${c}.err || { tail -5 $OUT/config${c}.err; exit 1; }
  echo "config $c done"
done
timeout -k 10 300 python tools/halo_bench.py --steps 10 --warmup 2 > $OUT/halo_1rank_512.json 2> $OUT/halo_1rank.err || { tail -5 $OUT/halo_1rank.err; exit 1; }
echo "halo one rank done"
timeout -k 10 300 python tools/halo_bench.py --steps 10 --warmup 2 --force-dist > $OUT/halo_1rank_rccl_self_512.json 2> $OUT/halo_1rank_rccl.err; echo "halo rccl self rc $?"; tail -3 $OUT/halo_1rank_rccl.err
timeout -k 10 300 python bench.py --simultaneous --no-cpu-baseline --no-abi > $OUT/simultaneous_1rank_bench.json 2>> $OUT/bench.err
echo "all done"
