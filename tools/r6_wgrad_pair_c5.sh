for v in 1 2 4; do
MFP_WGRAD_PAIR=$v python bench.py --config c5 --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c5 blocks per launch=$v', round(d['ms_per_step'],4), round(d['ms_per_step_median'],4), d['final_loss'])"
done
