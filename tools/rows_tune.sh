#!/bin/bash
# per-length tuning of the last-axis tile shape: the default build and variant builds (lib/v_s2.so,
# v_s4.so, v_s8.so = walking tiles of 2 / 4 / 8 rows for every length) over a ladder of d2
cd "$(dirname "$0")/.."
L=libmultiviewnative_amd/lib
cp $L/libmultiviewnative.so /tmp/_A.so
# ROWS_TUNE_CUBES=1: cubes d2^3 instead of 128 x 512 x d2
for d2 in ${ROWS_TUNE_D2:-192 256 320 384 512 576 640 768 960 1024 1280 1536 1920 2048}; do
  for v in default "$@"; do
    if [ "$v" != default ]; then cp $L/$v $L/libmultiviewnative.so; fi
    printf "d2 %5d %-10s " $d2 $v
    if [ -n "$ROWS_TUNE_CUBES" ]; then SH="$d2 $d2 $d2"; else SH="128 512 $d2"; fi
    MVN_NO_WAVE_ROWS=1 AB_NO_FFT=1 AB_SHAPE="$SH" python tools/sweep.py "" 2>&1 | grep view-iter | python -c "
import sys,ast
l=sys.stdin.read()
d=ast.literal_eval(l[l.index('{',3):]) if l else {}
print(' '.join('%s %.4f'%(k,d[k]) for k in ('rows_r2c','rows_c2r','rows_fused_div','rows_fused_upd') if k in d))"
    cp /tmp/_A.so $L/libmultiviewnative.so
  done
done
