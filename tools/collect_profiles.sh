#!/bin/bash
# Collects the evidence committed under profiles/ in ONE GPU-box call:
#   gpurun --timeout 1100 -- 'bash tools/collect_profiles.sh r1'
# then copy gpurun_out/prof_r1/* to profiles/r1_*.  Every GPU step has its own time limit; the first
# failure ends the script (no step runs after a fault or a timeout).
set -e -o pipefail
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
S=/tmp/prof_scratch_$$ && mkdir -p $S

echo "[1/6] bench line"
timeout -k 10 420 python3 "$R/bench.py" > "$O/bench_line.json" 2> "$O/bench_line.err"

echo "[2/6] kernel stats, inference legs (fp32 + bf16)"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $S/infer -o p -- \
    python3 "$R/bench.py" --mode infer --no-cpu-baseline > /dev/null 2> "$O/infer_stats.err"
cp "$(find $S/infer -name '*kernel_stats.csv' | head -1)" "$O/bench_infer_kernel_stats.csv"

echo "[3/6] kernel stats, fp32 training legs (mean/std + Gram style loss)"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $S/train -o p -- \
    python3 "$R/bench.py" --mode train --no-bf16 --no-cpu-baseline > /dev/null 2> "$O/train_stats.err"
cp "$(find $S/train -name '*kernel_stats.csv' | head -1)" "$O/train_step_kernel_stats.csv"

echo "[4/6] kernel stats, bf16 training leg (batch 8 at 512x512)"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $S/trainbf -o p -- \
    python3 "$R/bench.py" --mode train --compute bf16 --train-side 512 --no-gram --no-bf16 --no-cpu-baseline \
    > /dev/null 2> "$O/train_bf16_stats.err"
cp "$(find $S/trainbf -name '*kernel_stats.csv' | head -1)" "$O/train_bf16_512_kernel_stats.csv"

# the two PMC passes run the headline shape only (fp32 + bf16 legs at B=16, 512^2; no 2048^2 leg, whose launches
# of the same instantiations carry other byte counts and would move the per-launch average)
echo "[5/6] PMC pass: FETCH_SIZE"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $S/fetch -o p -- \
    python3 "$R/bench.py" --mode infer --no-cpu-baseline --no-hires --no-aux --steps 3 --warmup 1 > /dev/null 2> "$O/pmc_fetch.err"
echo "[6/6] PMC pass: WRITE_SIZE"
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $S/write -o p -- \
    python3 "$R/bench.py" --mode infer --no-cpu-baseline --no-hires --no-aux --steps 3 --warmup 1 > /dev/null 2> "$O/pmc_write.err"
python3 "$R/tools/pmc_hbm_csv.py" $S/fetch $S/write > "$O/bench_infer_pmc_hbm.csv"

echo "[7] loss kernels (histogram, TV, covariance, mean/std, MSE): kernel stats + FETCH_SIZE / WRITE_SIZE passes"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $S/aux -o p -- \
    python3 "$R/tools/bench_aux_kernels.py" > "$O/aux_kernels.txt" 2> "$O/aux.err"
cp "$(find $S/aux -name '*kernel_stats.csv' | head -1)" "$O/aux_kernel_stats.csv"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $S/auxf -o p -- \
    python3 "$R/tools/bench_aux_kernels.py" --no-graph > /dev/null 2> "$O/aux_fetch.err"
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $S/auxw -o p -- \
    python3 "$R/tools/bench_aux_kernels.py" --no-graph > /dev/null 2> "$O/aux_write.err"
python3 "$R/tools/pmc_hbm_csv.py" $S/auxf $S/auxw > "$O/aux_pmc_hbm.csv"

echo "[8] kernel stats, Gram-variant training step"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $S/gram -o p -- \
    python3 "$R/tools/gram_step.py" 12 > /dev/null 2> "$O/gram.err"
cp "$(find $S/gram -name '*kernel_stats.csv' | head -1)" "$O/train_gram_kernel_stats.csv"
echo "[9] small convs, AdaIN, augmentation, per-layer weight gradients: timing tables"
timeout -k 10 120 python3 "$R/tools/bench_small_convs.py" > "$O/small_convs.txt" 2> "$O/small.err"
timeout -k 10 120 python3 "$R/tools/bench_adain.py" > "$O/adain.txt" 2> "$O/adain.err"
timeout -k 10 120 python3 "$R/tools/bench_augment.py" > "$O/augment.txt" 2> "$O/augment.err"
timeout -k 10 120 python3 "$R/tools/bench_conv_layers.py" --batch 8 --mode wgrad --compute bf16 --iters 10 \
    > "$O/wgrad_bf16_layers.txt" 2> "$O/wgrad.err"
rm -rf $S
rm -f "$O"/*.err
echo done
