#!/bin/bash
# One GPU-box session for a round's closing records: tools/profile_round.sh <tag>, then the default bench.py run.
#   bash tools/final_records.sh <tag>      (from the repository root on the GPU box)
TAG=${1:-rXX}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
bash tools/profile_round.sh $TAG > gpurun_out/$TAG/profile_round.log 2>&1 || { tail -5 gpurun_out/$TAG/profile_round.log; exit 1; }
python bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err || { tail -5 gpurun_out/$TAG/bench.err; exit 1; }
tail -3 gpurun_out/$TAG/profile_round.log
python - <<PY
import json
d = json.load(open("gpurun_out/$TAG/bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["cpu_baseline"])
PY
head -6 gpurun_out/$TAG/kernel_stats.csv
