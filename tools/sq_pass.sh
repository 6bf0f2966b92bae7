#!/bin/bash
# SQ counter pass over tools/pmc_probe.py (LDS conflicts, LDS issue stalls, VALU count); extra
# arguments = environment settings (VAR=value) for the probe.   tools/sq_pass.sh <tag> [VAR=value ...]
set -e
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/sq -o p -- python3 $ROOT/tools/pmc_probe.py $PROBE_SHAPE > $OUT/sq.log 2>&1
cd $ROOT
python3 tools/rocpd_stats.py counters $OUT/sq_counters.md "SQ counters of the hot kernels (512^3, tools/pmc_probe.py) $*" $OUT/sq | grep "kw_rows\|kx_rows\|kx_strided\|^| kernel"
