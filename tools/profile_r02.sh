# usage (on the GPU box): bash tools/profile_r02.sh <tag> [extra bench flags]
# rocprofv3 --kernel-trace --stats of the default bench command -> gpurun_out/r02_<tag>_kernel_stats.csv
set -e
TAG=$1; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
STEPS=50; WARM=5
rocprofv3 --kernel-trace --stats -d $OUT -o t -- python bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --no-roofline "$@" > $OUT/bench.log 2>&1 || true
DB=$(ls $OUT/*.db $OUT/*/*.db 2>/dev/null | head -1)
# warmup steps + the 2 capture warm-up steps + capture itself run the kernels too: count launches of the
# step-counter kernel instead of trusting STEPS
python tools/rocprof_summary.py $DB gpurun_out/r02_${TAG}_kernel_stats.csv $((STEPS + WARM + 2))
python tools/step_timeline.py $DB > gpurun_out/r02_${TAG}_step_timeline.txt 2>/dev/null || true
tail -1 $OUT/bench.log > gpurun_out/r02_${TAG}_bench_under_profiler.json
head -30 gpurun_out/r02_${TAG}_kernel_stats.csv
