#!/bin/bash
# rocprofv3 kernel stats of the bf16 training leg (batch 8 at 512x512, eager): gpurun -- 'bash tools/prof_bf16_train.sh TAG'
set -e -o pipefail
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
S=/tmp/prof_scratch_$$ && mkdir -p $S
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $S/trainbf -o p -- \
    python3 "$R/bench.py" --mode train --compute bf16 --train-side 512 --no-gram --no-bf16 --no-cpu-baseline --no-hip-graph \
    > "$O/prof_${TAG}_line.json" 2> "$O/prof_${TAG}.err"
cp "$(find $S/trainbf -name '*kernel_stats.csv' | head -1)" "$O/prof_${TAG}_train_bf16_512_kernel_stats.csv"
