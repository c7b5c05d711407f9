#!/bin/bash
# HBM traffic of one train step from the L2 memory-side counters (MI355X_MICROARCH.md "HBM"):
# two separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit in one), kernel-trace only.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_step
mkdir -p $OUT
STEPS=${STEPS:-6}      # BENCH_ARGS="--config c5 --dtype bf16": another configuration
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o pmc -- \
    python bench.py --steps $STEPS --warmup 1 --no-cpu-baseline --no-roofline $BENCH_ARGS > $OUT/$c.log 2>&1 || true
done
python tools/pmc_step_summary.py $OUT $((2 * STEPS + 3)) | tee $OUT/summary.txt   # 2 capture warm-ups + 1 warm-up + K timed + K median pass
