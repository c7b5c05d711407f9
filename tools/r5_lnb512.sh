# LayerNorm backward in the epilogue of the d_model-512 input-gradient products (mfp_dense_n512_lnb): kernel test, c5 parity, same-box A/B
# bash tools/r5_lnb512.sh > gpurun_out/r05_lnb512.txt 2>&1   (GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { python bench.py $@ --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms  %.3f M elements/s' % (d['ms_per_step'], d['value'] / 1e6))"; }
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "dense_n512" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -k "c5" 2>&1 | tail -5
for rep in 1 2; do
  echo "c5 bf16 ln_bwd stand-alone: $(MFP_D512_LN_BWD=0 run --config c5 --dtype bf16 --steps 30 --warmup 5)"
  echo "c5 bf16 ln_bwd in os512   : $(MFP_D512_LN_BWD=1 run --config c5 --dtype bf16 --steps 30 --warmup 5)"
done
