for rep in 1 2 3; do
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('base  ', round(d['ms_per_step'],4))"
MFP_HIP_LIB=$PWD/tools/abl/libmfp_block_attn_1.so python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('prio 1', round(d['ms_per_step'],4))"
done
