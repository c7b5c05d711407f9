# same-box A/B of MFP_WGRAD_PAIR (blocks per grouped weight-gradient launch): 1 = one launch per block (round 5), 2, 4
for rep in 1 2; do
for v in 1 2 4; do
MFP_WGRAD_PAIR=$v python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c2 blocks per launch=$v', round(d['ms_per_step'],4), round(d['ms_per_step_median'],4), d['final_loss'])"
done; done
for v in 1 2 4; do
MFP_WGRAD_PAIR=$v python bench.py --config c4 --steps 200 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c4 blocks per launch=$v', round(d['ms_per_step'],4), round(d['ms_per_step_median'],4), d['final_loss'])"
done
