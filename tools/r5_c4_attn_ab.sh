cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { python bench.py $@ --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms  %.3f M elements/s' % (d['ms_per_step'], d['value'] / 1e6))"; }
for rep in 1 2; do
echo "c4 three launches    : $(run --config c4 --steps 100 --warmup 10)"
echo "c4 ATTN_BLOCK_BWD=1  : $(MFP_ATTN_BLOCK_BWD=1 run --config c4 --steps 100 --warmup 10)"
done
echo "seq64x256 three launches: $(run --config c2 --seq 64 --batch 256 --steps 100 --warmup 10)"
echo "seq64x256 one launch    : $(MFP_ATTN_BLOCK_BWD=1 run --config c2 --seq 64 --batch 256 --steps 100 --warmup 10)"
