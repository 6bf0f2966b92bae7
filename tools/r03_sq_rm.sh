#!/bin/bash
# SQ counter passes over tools/pmc_probe.py at 64 x 1920 x 1920 for A/B builds of the HIP library:
#   tools/r03_sq_rm.sh libmvn_ab_x.so libmvn_ab_y.so      (files under libmultiviewnative_amd/lib, MVN_PRODUCT_SO)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r03sqrm
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
s="64 1920 1920"
for v in "$@"; do
  export MVN_PRODUCT_SO=$ROOT/libmultiviewnative_amd/lib/$v
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/sq_$v -o p -- python3 $ROOT/tools/pmc_probe.py $s > $OUT/sq_$v.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $OUT/sq2_$v -o p -- python3 $ROOT/tools/pmc_probe.py $s > $OUT/sq2_$v.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_WAVES_EQ_64 -d $OUT/sq3_$v -o p -- python3 $ROOT/tools/pmc_probe.py $s > $OUT/sq3_$v.log 2>&1 || true
  cd $ROOT
  python3 tools/rocpd_stats.py counters $OUT/sq_$v.md "SQ counters $s $v" $OUT/sq_$v $OUT/sq2_$v $OUT/sq3_$v > /dev/null
  rm -rf $OUT/sq_$v $OUT/sq2_$v $OUT/sq3_$v
  cd /tmp
done
echo done
