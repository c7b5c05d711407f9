# LayerNorm-backward epilogues with 16-byte accesses (ln_bwd_tile16): kernel tests, model parity, same-box timing via git stash is not possible on the box -- compare with the numbers of the previous run
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { python bench.py $@ --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms  %.3f M elements/s' % (d['ms_per_step'], d['value'] / 1e6))"; }
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "mlp_bwd_ln or attn_block_bwd or dgrad_qkv_ln" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -k "timed_shape_parity or train_step_timed or seq64 or half" 2>&1 | tail -4
T=32768 WHICH=mlpbwd,attnbwd timeout 300 python tools/bench_fused.py 2>&1 | tail -3
for rep in 1 2; do
  echo "c2: $(run --config c2 --steps 100 --warmup 10)"
  echo "c4: $(run --config c4 --steps 100 --warmup 10)"
done
echo "seq64/512: $(run --config c2 --seq 64 --batch 512 --steps 100 --warmup 10)"
