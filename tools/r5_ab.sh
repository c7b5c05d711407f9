# A/B of one environment switch at a bench config, same box: bash tools/r5_ab.sh ENVVAR "bench args"   (GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
V=$1; shift
for rep in 1 2; do for v in 1 0; do
  env $V=$v python bench.py $@ --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$V=$v', round(d['ms_per_step'],4), round(d['ms_per_step_median'],4))"
done; done
OUT=gpurun_out/prof_ab; rm -rf $OUT; mkdir -p $OUT
for v in 1 0; do
rocprofv3 --kernel-trace --stats -d $OUT -o t$v -- env $V=$v python bench.py $@ --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $OUT/bench.log 2>&1 || true
DB=$(ls $OUT/*t$v*.db $OUT/*/*t$v*.db 2>/dev/null | head -1)
python tools/rocprof_summary.py $DB gpurun_out/ab_$v.csv 25 >/dev/null; echo "== $V=$v"; head -12 gpurun_out/ab_$v.csv | cut -c1-150
done
rm -rf $OUT
