mkdir -p gpurun_out/r5
for lib in default tools/abl/libmfp_block_attn_max-memory-clause.so tools/abl/libmfp_heads_loss_max-ilp.so tools/abl/libmfp_heads_loss_max-memory-clause.so tools/abl/libmfp_heads_loss_iterative-ilp.so tools/abl/libmfp_layernorm_max-ilp.so tools/abl/libmfp_layernorm_iterative-ilp.so tools/abl/libmfp_gemm_max-ilp.so tools/abl/libmfp_gemm_max-memory-clause.so tools/abl/libmfp_masking_max-ilp.so default; do
  if [ "$lib" = default ]; then unset MFP_HIP_LIB; else export MFP_HIP_LIB=$PWD/$lib; fi
  echo "== $lib" >> gpurun_out/r5/sweep.log
  python bench.py --steps 40 --warmup 5 --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['ms_per_step_median'])" >> gpurun_out/r5/sweep.log
done
