# Half-document tiles of the one-launch block forward (mfp_block_fwd_xhat_half): parity, stand-alone timing, c4 same-box A/B.
# bash tools/r5_half.sh > gpurun_out/r05_half_tiles.txt 2>&1   (GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { python bench.py $@ --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms  %.3f M elements/s' % (d['ms_per_step'], d['value'] / 1e6))"; }
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "test_block_fwd" 2>&1 | tail -5
T=16384 WHICH=block timeout 300 python tools/bench_fused.py 2>&1 | tail -6

for rep in 1 2; do
  echo "c4 full tiles: $(MFP_BLOCK_HALF=0 run --config c4 --steps 100 --warmup 10)"
  echo "c4 half tiles 4 waves: $(MFP_BLOCK_HALF=1 MFP_BLOCK_HALF_WAVES=4 run --config c4 --steps 100 --warmup 10)"
  echo "c4 half tiles 8 waves: $(MFP_BLOCK_HALF=1 MFP_BLOCK_HALF_WAVES=8 run --config c4 --steps 100 --warmup 10)"
done
echo "c2 full tiles: $(MFP_BLOCK_HALF=0 run --config c2 --steps 100 --warmup 10)"
echo "c2 half tiles: $(MFP_BLOCK_HALF=1 run --config c2 --steps 100 --warmup 10)"
