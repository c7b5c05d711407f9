set -x
mkdir -p gpurun_out/r5
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r5/g1_pytest.log
python tools/torch_ops_in_step.py > gpurun_out/r5/g1_torch_ops.log 2>&1
CAPTURE=1 python tools/torch_ops_in_step.py > gpurun_out/r5/g1_torch_ops_capture.log 2>&1
tools/ubench/prod > gpurun_out/r5/g1_prod.log 2>&1
WHICH=block python tools/bench_fused.py > gpurun_out/r5/g1_block.log 2>&1
python bench.py --steps 40 --warmup 5 > gpurun_out/r5/g1_c2.json 2> gpurun_out/r5/g1_c2.err
python bench.py --config c5 --dtype bf16 --steps 30 --warmup 5 > gpurun_out/r5/g1_c5.json 2> gpurun_out/r5/g1_c5.err
