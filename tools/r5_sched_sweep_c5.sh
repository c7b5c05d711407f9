mkdir -p gpurun_out/r5
for lib in default tools/abl/libmfp_block_d512_max-memory-clause.so tools/abl/libmfp_attention_max-ilp.so tools/abl/libmfp_attention_max-memory-clause.so tools/abl/libmfp_layernorm_max-memory-clause.so default; do
  if [ "$lib" = default ]; then unset MFP_HIP_LIB; else export MFP_HIP_LIB=$PWD/$lib; fi
  echo "== $lib" >> gpurun_out/r5/sweep_c5.log
  python bench.py --config c5 --dtype bf16 --steps 30 --warmup 5 --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['ms_per_step_median'])" >> gpurun_out/r5/sweep_c5.log
done
