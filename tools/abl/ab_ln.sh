# LayerNorm-backward prefetch depth sweep (tools/abl/build_abl.sh layernorm LN_BWD_DEPTH 1 3 4): bash tools/abl/ab_ln.sh on the GPU box.
# Round 5: depth 1 / 2 (shipped) / 3 / 4 -> c5 23.2 / 23.5 / 30.6 / 29.0 us, c4 11.9 / 12.4 / 12.9 / 13.9 us per launch: not a latency chain.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in flex-dm_amd/mfp/hip/libmfp_hip.so tools/abl/libmfp_layernorm_1.so tools/abl/libmfp_layernorm_3.so tools/abl/libmfp_layernorm_4.so; do
for cfg in "--config c5 --dtype bf16" "--config c4"; do
OUT=gpurun_out/prof_ab; rm -rf $OUT; mkdir -p $OUT
MFP_HIP_LIB=$GRAFT_REPO_ROOT/$lib rocprofv3 --kernel-trace --stats -d $OUT -o t -- python bench.py $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $OUT/bench.log 2>&1
DB=$(ls $OUT/*.db $OUT/*/*.db 2>/dev/null | head -1); python tools/rocprof_summary.py $DB gpurun_out/ab.csv 25 >/dev/null; echo "== $lib $cfg: $(grep -E "ln_bwd" gpurun_out/ab.csv | cut -d, -f1,4 | tr '\n' ' ')"; rm -rf $OUT
done; done
