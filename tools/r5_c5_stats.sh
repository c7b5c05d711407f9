# rocprofv3 kernel stats of the c5 bf16 step -> gpurun_out/r05_kernel_stats_c5_bf16.csv   (GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out; steps=20
OUT=$O/prof_cfg; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT -o t -- python bench.py --config c5 --dtype bf16 --steps $steps --warmup 5 --no-cpu-baseline --no-roofline > $OUT/bench.log 2>&1 || true
DB=$(ls $OUT/*.db $OUT/*/*.db 2>/dev/null | head -1)
python tools/rocprof_summary.py $DB $O/r05_kernel_stats_c5_bf16.csv $((2 * steps + 5 + 2))
rm -rf $OUT
head -24 $O/r05_kernel_stats_c5_bf16.csv
