# mask_kernel: sixteen tokens per workgroup vs the wave-per-token form -- bit-identical outputs, oracle masking parity, c2 / c4 same-box A/B
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { python bench.py $@ --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms  %.3f M elements/s' % (d['ms_per_step'], d['value'] / 1e6))"; }
timeout 900 python -m pytest tests/test_gpu_callers.py -q -x -k "mask" 2>&1 | tail -4
for rep in 1 2; do
  echo "c2 mask wave-per-token : $(MFP_MASK_WAVE=1 run --config c2 --steps 100 --warmup 10)"
  echo "c2 mask 16 tokens / wg : $(run --config c2 --steps 100 --warmup 10)"
done
echo "c3 mask wave-per-token : $(MFP_MASK_WAVE=1 run --config c3 --steps 100 --warmup 10)"
echo "c3 mask 16 tokens / wg : $(run --config c3 --steps 100 --warmup 10)"
echo "c4 mask wave-per-token : $(MFP_MASK_WAVE=1 run --config c4 --steps 100 --warmup 10)"
echo "c4 mask 16 tokens / wg : $(run --config c4 --steps 100 --warmup 10)"
