#!/bin/bash
# Step A1 of the L2-residency probe (profiles/r03_l2_probe.md): per-kernel times of one view update at
# 512^3 with the -DMVN_PROBE build (`make probe`), tiles wrapped onto a working set that stays in the
# XCD's 4 MB L2.  Results of the wrapped runs are garbage by construction; only the times count.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r03
mkdir -p $OUT
export MVN_PRODUCT_SO=$PWD/libmultiviewnative_amd/lib/libmultiviewnative_probe.so
export AB_NO_FFT=1
for cfg in "0 0" "8 64" "16 128" "32 256" "64 512" "256 4096" "0 0"; do
  set -- $cfg
  echo "== MVN_PROBE_WRAP_ST=$1 (x 64 KB tiles) MVN_PROBE_WRAP_ROWS=$2 (x 4 KB row pairs)"
  MVN_PROBE_WRAP_ST=$1 MVN_PROBE_WRAP_ROWS=$2 timeout -k 10 120 python tools/sweep.py "" 2>&1 | grep view-iter || exit 1
done
