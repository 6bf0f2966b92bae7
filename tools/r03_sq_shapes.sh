#!/bin/bash
# SQ counters and launch statistics of the hot kernels at the non-power-of-two shapes (H = 960, H = 288)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r03sq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for s in "64 1920 1920" "576 576 576" "512 512 512"; do
  t=$(echo $s | tr ' ' 'x')
  rocprofv3 --kernel-trace --stats -d $OUT/st_$t -o p -- python3 $ROOT/tools/pmc_probe.py $s > $OUT/st_$t.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/sq_$t -o p -- python3 $ROOT/tools/pmc_probe.py $s > $OUT/sq_$t.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_INST_CYCLES_VMEM -d $OUT/sq2_$t -o p -- python3 $ROOT/tools/pmc_probe.py $s > $OUT/sq2_$t.log 2>&1
  cd $ROOT
  python3 tools/rocpd_stats.py stats $OUT/st_$t $OUT/kernel_stats_$t.csv > /dev/null
  python3 tools/rocpd_stats.py counters $OUT/sq_$t.md "SQ counters $s" $OUT/sq_$t $OUT/sq2_$t > /dev/null
  rm -rf $OUT/st_$t $OUT/sq_$t $OUT/sq2_$t
  cd /tmp
  echo "$s done"
done
