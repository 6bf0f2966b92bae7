#!/bin/bash
# Step A2 of the L2-residency probe: tools/l2_plane_probe.hip timed, then its HBM traffic per kernel
# (separate FETCH_SIZE / WRITE_SIZE passes, program directly after `--`).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r03
mkdir -p $OUT
hipcc -O3 --offload-arch=gfx950 -o /tmp/l2_plane_probe $ROOT/tools/l2_plane_probe.hip
timeout -k 10 300 /tmp/l2_plane_probe > $OUT/a2_times.txt 2>&1
echo "timing done"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/a2_rd -o p --output-format csv -- /tmp/l2_plane_probe pmc > $OUT/a2_rd.log 2>&1
echo "fetch pass done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/a2_wr -o p --output-format csv -- /tmp/l2_plane_probe pmc > $OUT/a2_wr.log 2>&1
echo "write pass done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $OUT/a2_tcc -o p --output-format csv -- /tmp/l2_plane_probe pmc > $OUT/a2_tcc.log 2>&1 || echo "tcc pass failed"
cd $ROOT
python3 tools/l2_probe_summarize.py $OUT > $OUT/a2_traffic.md
cat $OUT/a2_traffic.md
