set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof_k
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_k -o r01k -- python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/prof_k/bench.log 2>&1 || true
DB=$(ls gpurun_out/prof_k/*.db gpurun_out/prof_k/*/*.db 2>/dev/null | head -1)
echo DB=$DB
python tools/rocprof_summary.py $DB gpurun_out/r01_k_kernel_stats.csv 57
python tools/step_timeline.py $DB > gpurun_out/r01_k_step_timeline.txt
STEPS=6 bash tools/pmc_step.sh > gpurun_out/pmc_step_k.log 2>&1 || true
python bench.py --steps 100 --warmup 10 2>/dev/null | tail -1 > gpurun_out/r01_k_bench.json
head -12 gpurun_out/r01_k_kernel_stats.csv; cat gpurun_out/r01_k_step_timeline.txt | head -20; tail -5 gpurun_out/pmc_step_k.log
