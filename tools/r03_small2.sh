#!/bin/bash
# product defaults (direct dim0 leg at every size): differential fuzz against the oracle, piece-target A/B,
# block sequences through the ABI
set -e
O=gpurun_out/r03s2
mkdir -p $O
FUZZ_DEEP=1 timeout -k 10 400 python3 tools/fuzz_shapes.py 70 11 > $O/fuzz_deep.txt 2>&1
tail -2 $O/fuzz_deep.txt
FUZZ_DEEP=1 FUZZ_PAD=zero timeout -k 10 400 python3 tools/fuzz_shapes.py 40 12 > $O/fuzz_deep_zero.txt 2>&1
tail -2 $O/fuzz_deep_zero.txt
FUZZ_FIXED=1 FUZZ_DEEP=1 FUZZ_MAX_VOXELS=40000000 timeout -k 10 400 python3 tools/fuzz_shapes.py 24 13 > $O/fuzz_fixed.txt 2>&1
tail -2 $O/fuzz_fixed.txt
export AB_NO_FFT=1
for s in "256 256 256" "192 192 192" "128 256 256" "320 320 320"; do
  echo "== $s" >> $O/piece_target.txt
  AB_SHAPE="$s" python3 tools/sweep.py "MVN_DIM0_DIRECT_MIN_PLANE=65536" "MVN_DIM0_DIRECT_MIN_PLANE=98304" "MVN_DIM0_DIRECT_MIN_PLANE=131072" "MVN_DIM0_DIRECT_MIN_PLANE=196608" "MVN_DIM0_DIRECT_MIN_PLANE=65536" "MVN_DIM0_DIRECT_MIN_PLANE=98304" >> $O/piece_target.txt 2>&1
done
for e in 64 128 256; do
  python3 tools/abi_end_to_end.py $e 8 zero >> $O/abi_blocks.txt 2>&1
  MVN_DIM0_DIRECT=0 python3 tools/abi_end_to_end.py $e 8 zero >> $O/abi_blocks_fft_leg.txt 2>&1
done
cat $O/abi_blocks.txt
