# Round-2 evidence, run on the GPU box: bash tools/profile_round2.sh
#   1. rocprofv3 --kernel-trace --stats of the default bench command -> r02_kernel_stats.csv, r02_step_dump.txt
#   2. HBM traffic per kernel (separate --pmc FETCH_SIZE / WRITE_SIZE passes) -> r02_pmc_traffic.{json,_summary.txt}
#   3. the bench lines (c2 with roofline.traffic from step 2, c3, c5 fp8 and bf16)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
bash tools/profile_r02.sh final > /dev/null 2>&1 || true
cp $O/r02_final_kernel_stats.csv $O/r02_kernel_stats.csv
python tools/step_dump.py $O/prof_final/t_results.db > $O/r02_step_dump.txt
STEPS=6 bash tools/pmc_step.sh > $O/r02_pmc_step.log 2>&1 || true
cp $O/pmc_step/pmc_traffic.json $O/r02_pmc_traffic.json
cp $O/pmc_step/summary.txt $O/r02_pmc_traffic_summary.txt
mkdir -p profiles && cp $O/r02_pmc_traffic.json profiles/r02_pmc_traffic.json     # bench.py reads it from profiles/
python bench.py --steps 100 --warmup 10 2>/dev/null | tail -1 > $O/r02_bench_c2.json
python bench.py --steps 100 --warmup 10 --config c3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r02_bench_c3.json
python bench.py --steps 30 --warmup 5 --config c5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r02_bench_c5_fp8.json
python bench.py --steps 30 --warmup 5 --config c5 --dtype bf16 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r02_bench_c5_bf16.json
python bench.py --steps 100 --warmup 10 --config c4 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/r02_bench_c4_1gpu.json
head -40 $O/r02_kernel_stats.csv; tail -3 $O/r02_pmc_traffic_summary.txt; python -c "
import json
for n in ('c2','c3','c5_fp8','c5_bf16','c4_1gpu'):
    d=json.load(open('$O/r02_bench_%s.json'%n)); print(n, round(d['ms_per_step'],3), round(d['value']))
"
