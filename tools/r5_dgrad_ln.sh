# LN1 backward inside dgrad_half_kernel<768> (mfp_dgrad_qkv_ln_half): kernel test, half-route oracle parity, c4 same-box A/B
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { python bench.py $@ --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms  %.3f M elements/s' % (d['ms_per_step'], d['value'] / 1e6))"; }
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "dgrad_qkv" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -k "half or timed_shape_parity" 2>&1 | tail -4
for rep in 1 2; do
  echo "c4 ln_bwd stand-alone      : $(MFP_DGRAD_LN_HALF=0 run --config c4 --steps 100 --warmup 10)"
  echo "c4 ln_bwd in dgrad_half<768>: $(MFP_DGRAD_LN_HALF=1 run --config c4 --steps 100 --warmup 10)"
done
