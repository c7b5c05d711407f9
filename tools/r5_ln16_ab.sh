# same-box A/B: LayerNorm-backward epilogues with 8-byte (tools/abl/libmfp_ln8.so: the previous commit's block_fused / block_attn_bwd) vs 16-byte accesses
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { python bench.py $@ --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms  %.3f M elements/s' % (d['ms_per_step'], d['value'] / 1e6))"; }
for rep in 1 2; do
  echo "c2  8-byte LN epilogues: $(MFP_HIP_LIB=tools/abl/libmfp_ln8.so run --config c2 --steps 100 --warmup 10)"
  echo "c2 16-byte LN epilogues: $(run --config c2 --steps 100 --warmup 10)"
done
echo "c4  8-byte: $(MFP_HIP_LIB=tools/abl/libmfp_ln8.so run --config c4 --steps 100 --warmup 10)"
echo "c4 16-byte: $(run --config c4 --steps 100 --warmup 10)"
echo "seq64/512  8-byte: $(MFP_HIP_LIB=tools/abl/libmfp_ln8.so run --config c2 --seq 64 --batch 512 --steps 100 --warmup 10)"
echo "seq64/512 16-byte: $(run --config c2 --seq 64 --batch 512 --steps 100 --warmup 10)"
