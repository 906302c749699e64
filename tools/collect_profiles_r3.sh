#!/bin/bash
# Round-3 additions to tools/collect_profiles.sh (same rules: one GPU-box call, every GPU step under its own limit,
# the first failure ends the script):  gpurun --timeout 900 -- 'bash tools/collect_profiles_r3.sh r3'
set -e -o pipefail
TAG=${1:-r3}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
S=/tmp/prof_scratch_r3_$$ && mkdir -p $S
TRAIN_BF16="--mode train --compute bf16 --train-side 512 --no-gram --no-bf16 --no-cpu-baseline --no-hip-graph"

echo "[1] PMC pass, bf16 training leg: MFMA busy + clock per kernel"
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv \
    -d $S/mf -o p -- python3 "$R/bench.py" $TRAIN_BF16 --steps 3 --warmup 1 > /dev/null 2> "$O/mf.err"
python3 "$R/tools/pmc_mfma_table.py" $S/mf conv3x3 wgrad dgrad_band > "$O/pmc_mfma_busy_train_bf16.txt"

echo "[2] PMC passes, bf16 training leg: FETCH_SIZE / WRITE_SIZE per kernel"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $S/tf -o p -- \
    python3 "$R/bench.py" $TRAIN_BF16 --steps 3 --warmup 1 > /dev/null 2> "$O/tf.err"
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $S/tw -o p -- \
    python3 "$R/bench.py" $TRAIN_BF16 --steps 3 --warmup 1 > /dev/null 2> "$O/tw.err"
python3 "$R/tools/pmc_hbm_csv.py" $S/tf $S/tw > "$O/train_bf16_pmc_hbm.csv"

echo "[3] per-layer tables: decoder chain data gradient (fused fold vs workspace + fold kernel), staged weight gradient, loss-VGG G8 launches"
timeout -k 10 120 python3 "$R/tools/bench_chain_dgrad.py" > "$O/chain_dgrad_layers.txt" 2> "$O/cd.err"
timeout -k 10 120 python3 "$R/tools/bench_wgrad_pre.py" > "$O/wgrad_pre_layers.txt" 2> "$O/wp.err"
ST_WGRAD_NO_XCD=1 timeout -k 10 120 python3 "$R/tools/bench_wgrad_pre.py" > "$O/wgrad_pre_layers_plain_grid.txt" 2> "$O/wpn.err"
timeout -k 10 120 python3 "$R/tools/bench_vgg_dgrad_gate.py" > "$O/vgg_g8_launches.txt" 2> "$O/vg.err"
rm -rf $S
rm -f "$O"/*.err
echo done
