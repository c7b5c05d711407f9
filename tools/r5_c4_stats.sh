# rocprofv3 kernel stats + bench line of the c4 per-GPU shape -> gpurun_out/r05_kernel_stats_c4_bf16.csv, r05_bench_c4_1gpu.json   (GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out; steps=50
OUT=$O/prof_cfg; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT -o t -- python bench.py --config c4 --dtype bf16 --steps $steps --warmup 5 --no-cpu-baseline --no-roofline > $OUT/bench.log 2>&1 || true
DB=$(ls $OUT/*.db $OUT/*/*.db 2>/dev/null | head -1)
python tools/rocprof_summary.py $DB $O/r05_kernel_stats_c4_bf16.csv $((2 * steps + 5 + 2))
rm -rf $OUT
python bench.py --steps 100 --warmup 10 --config c4 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r05_bench_c4_1gpu.json
head -18 $O/r05_kernel_stats_c4_bf16.csv; python -c "
import json; d=json.load(open('$O/r05_bench_c4_1gpu.json')); print(round(d['ms_per_step'],4), round(d['value']))"
