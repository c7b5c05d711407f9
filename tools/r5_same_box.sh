# Same-box before / after of the second half of round 5 (the environment switches restore the earlier paths):
#   c2: LayerNorm backward inside the input-gradient launches, x-hat stash   (MFP_MLP_BWD_LN, MFP_ATTN_BWD_LN, MFP_XHAT_STASH)
#   c4: + 16-row LayerNorm-backward workgroups                          (MFP_LN_BWD_ROWS)
#   c5: query-split attention forward, bf16 residual-gradient stream, 16-row LayerNorm-backward workgroups
# bash tools/r5_same_box.sh > gpurun_out/r05_same_box_ab.txt   (GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { python bench.py $@ --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms  %.3f M elements/s' % (d['ms_per_step'], d['value'] / 1e6))"; }
rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk" | head -2
for rep in 1 2; do
  echo "c2 before: $(MFP_MLP_BWD_LN=0 MFP_ATTN_BWD_LN=0 MFP_XHAT_STASH=0 run --config c2 --steps 100 --warmup 10)"
  echo "c2 after : $(run --config c2 --steps 100 --warmup 10)"
  echo "c4 before: $(MFP_MLP_BWD_LN=0 MFP_ATTN_BWD_LN=0 MFP_XHAT_STASH=0 MFP_LN_BWD_ROWS=32 run --config c4 --steps 100 --warmup 10)"
  echo "c4 after : $(run --config c4 --steps 100 --warmup 10)"
  echo "c5 bf16 before: $(MFP_ATTN_FWD_QSPLIT=0 MFP_RES_GRAD_BF16=0 MFP_LN_BWD_ROWS=32 run --config c5 --dtype bf16 --steps 30 --warmup 5)"
  echo "c5 bf16 after : $(run --config c5 --dtype bf16 --steps 30 --warmup 5)"
done
