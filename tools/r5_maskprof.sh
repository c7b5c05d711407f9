cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for lib in flex-dm_amd/mfp/hip/libmfp_hip.so tools/abl/libmfp_masking_1.so tools/abl/libmfp_masking_2.so tools/abl/libmfp_masking_3.so; do
  OUT=gpurun_out/prof_m; rm -rf $OUT; mkdir -p $OUT
  MFP_HIP_LIB=$lib rocprofv3 --kernel-trace --stats -d $OUT -o t -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $OUT/bench.log 2>&1 || true
  DB=$(ls $OUT/*.db $OUT/*/*.db 2>/dev/null | head -1)
  python tools/rocprof_summary.py $DB $OUT/stats.csv 47
  echo "$lib: $(grep -E "^mask" $OUT/stats.csv)"
  rm -rf $OUT
done
