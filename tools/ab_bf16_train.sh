#!/bin/bash
# same-box A/B of the bf16 training leg (batch 8, 512x512): bash tools/ab_bf16_train.sh "ENV_A=1" "ENV_B=1" ...
# prints img/s and ms/step for the default and for every environment setting given, twice each (A B A B) 
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
run() { env $1 timeout -k 10 200 python3 "$R/bench.py" --mode train --no-cpu-baseline --no-hip-graph --no-gram --no-bf16 --compute bf16 \
  --train-side 512 --train-steps 20 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
  run "ST_AB_DEFAULT=1" || exit 1
  for e in "$@"; do run "$e" || exit 1; done
done
