# same-box A/B: residual rows of the x1 epilogue requested at the head of the last attention chunk (default, round 6) vs in the epilogue (AB_RES_FULL=0)
for rep in 1 2 3; do
MFP_HIP_LIB=$PWD/tools/abl/libmfp_block_attn_1.so python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('last-chunk loads (AB_RES_FULL=1)', round(d['ms_per_step'],4), d['final_loss'])"
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('epilogue loads (default)', round(d['ms_per_step'],4), d['final_loss'])"
done
MFP_HIP_LIB=$PWD/tools/abl/libmfp_block_attn_1.so python bench.py --seq 64 --batch 512 --steps 100 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('S=64 epilogue loads', round(d['ms_per_step'],4))"
python bench.py --seq 64 --batch 512 --steps 100 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('S=64 last-chunk    ', round(d['ms_per_step'],4))"
python -m pytest tests/test_gpu_kernels.py -q -k "block_fwd or attn_block_fwd or block_infer" 2>&1 | tail -3
