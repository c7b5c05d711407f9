# k-slices of the heads / encoder weight-gradient groups (22 tiles each at c2: 176 workgroups at 8 slices, two fit a CU)
run() { python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline $2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4))"; }
for rep in 1 2; do
run "default         "
MFP_WGRAD_SPLITK_HEADS=16 run "heads 16        "
MFP_WGRAD_SPLITK_HEADS=24 run "heads 24        "
MFP_WGRAD_SPLITK_ENC=16 run "enc 16          "
MFP_WGRAD_SPLITK_ENC=24 run "enc 24          "
MFP_WGRAD_SPLITK_HEADS=16 MFP_WGRAD_SPLITK_ENC=16 run "heads 16 enc 16 "
done
