#!/bin/bash
# On the GPU box: run one command with each variant library in turn (default build first).
#   tools/run_variants.sh "<tags>" <command...>
tags=$1; shift
echo "== default"; timeout -k 10 120 "$@" || exit 1
cp style_transfer_amd/libst_hip.so /tmp/libst_hip_default.so
for t in $tags; do
  echo "== $t"; cp exp/libst_hip_$t.so style_transfer_amd/libst_hip.so
  timeout -k 10 120 "$@" || exit 1
done
cp /tmp/libst_hip_default.so style_transfer_amd/libst_hip.so
