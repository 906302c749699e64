#!/bin/bash
# same-box A/B of two builds of the library on the bf16 training leg: A = in-tree, B = fake_lib/libst_norows.so (scratch)
R=${GRAFT_REPO_ROOT:-/root/repo}
run() { timeout -k 10 200 python3 "$R/bench.py" --mode train --no-cpu-baseline --no-hip-graph --no-gram --no-bf16 --compute bf16 \
  --train-side 512 --train-steps 20 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
cp "$R/style_transfer_amd/libst_hip.so" /tmp/new.so
for rep in 1 2; do
  cp /tmp/new.so "$R/style_transfer_amd/libst_hip.so"; run new || exit 1
  cp "$R/fake_lib/libst_norows.so" "$R/style_transfer_amd/libst_hip.so"; run base || exit 1
done
cp /tmp/new.so "$R/style_transfer_amd/libst_hip.so"
