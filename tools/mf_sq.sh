#!/bin/bash
# SQ counter passes over the fused middle pass probe (build/probe/mfp_<variant>):  tools/mf_sq.sh <variant> [k]
V=$1; K=${2:-31}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r04f/sq_$V
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export MF_NOCHECK=1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/a -o p -- $ROOT/build/probe/mfp_$V 512 256 $K 0 5 1 > $OUT/a.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_ANY -d $OUT/b -o p -- $ROOT/build/probe/mfp_$V 512 256 $K 0 5 1 > $OUT/b.log 2>&1
cd $ROOT
python3 tools/rocpd_stats.py counters $OUT/a.md "fused middle pass probe $V k=$K (a)" $OUT/a | grep "kf_mid\|^| kernel"
python3 tools/rocpd_stats.py counters $OUT/b.md "fused middle pass probe $V k=$K (b)" $OUT/b | grep "kf_mid\|^| kernel"
