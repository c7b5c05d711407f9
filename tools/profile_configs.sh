# rocprofv3 --kernel-trace --stats of the other BASELINE configurations (c4's per-GPU share, c5 bf16 / fp8, c3):
#   bash tools/profile_configs.sh   (on the GPU box) -> gpurun_out/r04_kernel_stats_<config>.csv
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
for spec in "c3:bf16:50" "c4:bf16:50" "c5:bf16:20" "c5:fp8:20"; do
  IFS=: read cfg dt steps <<< "$spec"
  OUT=$O/prof_cfg; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats -d $OUT -o t -- python bench.py --config $cfg --dtype $dt --steps $steps --warmup 5 --no-cpu-baseline --no-roofline > $OUT/bench.log 2>&1 || true
  DB=$(ls $OUT/*.db $OUT/*/*.db 2>/dev/null | head -1)
  python tools/rocprof_summary.py $DB $O/r04_kernel_stats_${cfg}_${dt}.csv $((2 * steps + 5 + 2))
  rm -rf $OUT
  head -14 $O/r04_kernel_stats_${cfg}_${dt}.csv
done
