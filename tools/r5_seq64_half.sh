# S = 64 half tiles (one document per eight-wave workgroup): kernel test, oracle parity at S = 64, same-box A/B at the reference's
# default batch (256 documents x 64 positions = 128 two-document tiles)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { python bench.py $@ --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms  %.3f M elements/s' % (d['ms_per_step'], d['value'] / 1e6))"; }
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "two_documents_per_tile or test_block_fwd" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -k "seq64" 2>&1 | tail -4
for rep in 1 2; do
  echo "seq 64, 256 documents, whole tiles: $(MFP_BLOCK_HALF=0 MFP_MLP_BWD_HALF=0 run --config c2 --seq 64 --batch 256 --steps 100 --warmup 10)"
  echo "seq 64, 256 documents, half tiles : $(run --config c2 --seq 64 --batch 256 --steps 100 --warmup 10)"
done
echo "seq 64, 512 documents (full chip): $(run --config c2 --seq 64 --batch 512 --steps 100 --warmup 10)"
