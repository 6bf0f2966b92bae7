#!/bin/bash
# Memory-path counters of the hot kernels (profiles/r04_mem_counters*.md): one rocprofv3 --pmc pass per counter
# group over tools/pmc_probe.py, --kernel-trace only, program directly after `--`.
#   tools/r04_mem_counters.sh TAG D0 D1 D2      (run from anywhere on the GPU box; output gpurun_out/TAG/mem)
# No TA_* counters: the TA block's group hung rocprofv3 on this pool in round 3 (killed at its time limit,
# profiles/r03_mem_counters.md) - not retried.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:-r04}
shift
SHAPE="${*:-512 512 512}"
OUT=$ROOT/gpurun_out/$TAG/mem
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i + 1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $group -d $OUT/p$i -o p -- python3 $ROOT/tools/pmc_probe.py $SHAPE > $OUT/p$i.log 2>&1
  rc=$?
  echo "pass $i ($group): rc $rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out: stopping"; exit 1; fi
done <<'GROUPS'
TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_LEVEL_sum
TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum
TCC_EA0_WRREQ_LEVEL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_SRC_FIFO_FULL_sum TCC_LATENCY_FIFO_FULL_sum
TCC_BUSY_sum TCC_CYCLE_sum TCC_READ_sum TCC_WRITE_sum
SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE
SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE
GROUPS
cd $ROOT
python3 tools/mem_counters_md.py $OUT > $OUT/../mem_counters.md
echo "summary written"
