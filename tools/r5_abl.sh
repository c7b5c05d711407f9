mkdir -p gpurun_out/r5
for a in 0 1 2 3 4; do
  echo "===== D5_ABL=$a" >> gpurun_out/r5/abl.log
  MFP_HIP_LIB=$PWD/tools/abl/libmfp_block_d512_t$a.so WHICH=as python tools/trace_d512.py 2>/dev/null | grep -A8 "as512 LN1" >> gpurun_out/r5/abl.log
done
