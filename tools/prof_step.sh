# usage (on the GPU box): bash tools/prof_step.sh <tag> [extra bench flags]
# rocprofv3 --kernel-trace of the bench command; prints one replayed step kernel by kernel -> gpurun_out/<tag>_step_dump.txt
TAG=$1; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace -d $OUT -o t -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline "$@" > $OUT/bench.log 2>&1 || true
DB=$(ls $OUT/*.db $OUT/*/*.db 2>/dev/null | head -1)
python tools/step_dump.py $DB > gpurun_out/${TAG}_step_dump.txt
rm -rf $OUT
cat gpurun_out/${TAG}_step_dump.txt
