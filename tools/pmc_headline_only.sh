set -e -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r3; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
S=/tmp/pmc_only_$$ && mkdir -p $S
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $S/fetch -o p -- python3 "$R/bench.py" --mode infer --no-cpu-baseline --no-hires --no-aux --steps 3 --warmup 1 > /dev/null 2> "$O/pmc_fetch.err"
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $S/write -o p -- python3 "$R/bench.py" --mode infer --no-cpu-baseline --no-hires --no-aux --steps 3 --warmup 1 > /dev/null 2> "$O/pmc_write.err"
python3 "$R/tools/pmc_hbm_csv.py" $S/fetch $S/write > "$O/bench_infer_pmc_hbm.csv"
grep -E "f32_kernel<2, 2, 2, 2|bf16_kernel<2, 4, 2, 4, false, false, false, false, true, true" "$O/bench_infer_pmc_hbm.csv"
