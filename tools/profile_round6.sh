# Round-6 evidence, run on the GPU box: bash tools/profile_round6.sh   (-> gpurun_out/r06_*, copy into profiles/)
#   1. rocprofv3 --kernel-trace --stats of the default bench command -> r06_kernel_stats.csv, r06_step_dump.txt
#   2. HBM traffic per kernel (separate --pmc FETCH_SIZE / WRITE_SIZE passes) -> r06_pmc_traffic.{json,_summary.txt}
#   3. matrix-pipe / LDS counters of the shipped kernels -> r06_pmc_mfma_lds.txt
#   4. the bench lines (c2 with roofline + cpu_baseline, c3, c4, c5 bf16 / fp8)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
OUT=$O/prof_r06; rm -rf $OUT; mkdir -p $OUT
STEPS=50; WARM=5
rocprofv3 --kernel-trace --stats -d $OUT -o t -- python bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --no-roofline > $OUT/bench.log 2>&1 || true
DB=$(ls $OUT/*.db $OUT/*/*.db 2>/dev/null | head -1)
python tools/rocprof_summary.py $DB $O/r06_kernel_stats.csv $((2 * STEPS + WARM + 2))   # 2 capture warm-ups + W + K timed + K for the median pass
python tools/step_dump.py $DB > $O/r06_step_dump.txt
rm -rf $OUT
STEPS=6 bash tools/pmc_step.sh > $O/r06_pmc_step.log 2>&1 || true
cp $O/pmc_step/pmc_traffic.json $O/r06_pmc_traffic.json
cp $O/pmc_step/summary.txt $O/r06_pmc_traffic_summary.txt
rm -rf $O/pmc_step/FETCH_SIZE $O/pmc_step/WRITE_SIZE
mkdir -p profiles && cp $O/r06_pmc_traffic.json profiles/r06_pmc_traffic.json     # bench.py reads it from profiles/
TAG=r06 STEPS=6 bash tools/pmc_mfma.sh > $O/r06_pmc_mfma.log 2>&1 || true
rm -rf $O/pmc_mfma
python bench.py --steps 100 --warmup 10 2>/dev/null | tail -1 > $O/r06_bench_c2.json
python bench.py --steps 100 --warmup 10 --config c3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r06_bench_c3.json
python bench.py --steps 100 --warmup 10 --config c4 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r06_bench_c4_1gpu.json
python bench.py --steps 100 --warmup 10 --config c1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r06_bench_c1.json
python bench.py --steps 30 --warmup 5 --config c5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r06_bench_c5_bf16.json      # (the default: bf16, with fp8_mode beside it)
head -30 $O/r06_kernel_stats.csv; tail -3 $O/r06_pmc_traffic_summary.txt; tail -25 $O/r06_pmc_mfma_lds.txt; python -c "
import json
for n in ('c1','c2','c3','c4_1gpu','c5_bf16'):
    d=json.load(open('$O/r06_bench_%s.json'%n)); print(n, round(d['ms_per_step'],3), round(d['value']), (d.get('roofline') or {}).get('encoder_block',{}).get('mfma_frac'))
"
python bench.py --steps 100 --warmup 10 --seq 64 --batch 512 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r06_bench_c2_seq64.json
for spec in "c4:bf16:50" "c5:bf16:20"; do
  IFS=: read cfg dt steps <<< "$spec"
  OUT=$O/prof_cfg; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats -d $OUT -o t -- python bench.py --config $cfg --dtype $dt --steps $steps --warmup 5 --no-cpu-baseline --no-roofline > $OUT/bench.log 2>&1 || true
  DB=$(ls $OUT/*.db $OUT/*/*.db 2>/dev/null | head -1)
  python tools/rocprof_summary.py $DB $O/r06_kernel_stats_${cfg}_${dt}.csv $((2 * steps + 5 + 2))
  rm -rf $OUT
  head -16 $O/r06_kernel_stats_${cfg}_${dt}.csv
done
