#!/bin/bash
# One GPU-box session that produces the raw material of profiles/<tag>_*: kernel-trace statistics
# of the bench command, the SQ counter pass and the two HBM-traffic passes over tools/pmc_probe.py.
#   tools/profile_round.sh <tag>        (run from the repository root on the GPU box)
# Counter passes carry --kernel-trace only (no other trace domain), the program follows `--` directly.
set -e
TAG=${1:-rXX}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o p -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-side > $OUT/stats_bench.json 2> $OUT/stats.err
echo "stats pass done"
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/sq -o p -- python3 $ROOT/tools/pmc_probe.py > $OUT/sq.log 2>&1
echo "sq pass done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_rd -o p --output-format csv -- python3 $ROOT/tools/pmc_probe.py > $OUT/rd.log 2>&1
echo "fetch pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE GRBM_GUI_ACTIVE -d $OUT/pmc_wr -o p --output-format csv -- python3 $ROOT/tools/pmc_probe.py > $OUT/wr.log 2>&1
echo "write pass done"
cd $ROOT
python3 tools/rocpd_stats.py stats $OUT/stats $OUT/kernel_stats.csv > /dev/null
python3 tools/rocpd_stats.py counters $OUT/sq_counters.md "SQ counters of the hot kernels (512^3, tools/pmc_probe.py)" $OUT/sq > /dev/null
python3 tools/pmc_summarize.py $OUT/pmc_rd $OUT/pmc_wr $OUT/pmc_traffic.json $OUT/pmc_traffic.md > /dev/null
echo "summaries written to $OUT"
