#!/bin/bash
# rocprofv3 --kernel-trace --stats of one python command; stats CSV -> gpurun_out/prof_TAG_kernel_stats.csv
#   bash tools/prof_cmd.sh TAG tools/bench_conv_layers.py --mode wgrad ...
set -e -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
SCRIPT=$R/$1; shift
cd /tmp && export TMPDIR=/tmp
S=/tmp/prof_scratch_$$_$TAG && mkdir -p $S
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $S -o p -- python3 "$SCRIPT" "$@" > "$O/prof_${TAG}.out" 2> "$O/prof_${TAG}.err"
cp "$(find $S -name '*kernel_stats.csv' | head -1)" "$O/prof_${TAG}_kernel_stats.csv"
