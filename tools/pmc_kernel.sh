#!/bin/bash
# usage: PROF_SCRIPT=x.py KFILTER=name tools/pmc_kernel.sh <outdir>   (separate --pmc passes, kernel-trace only: gpurun rule)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/tools/${PROF_SCRIPT:-prof_one_wg.py} > $OUT/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
agg = collections.OrderedDict()
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "${KFILTER:-attn_bwd}" not in r["Kernel_Name"]: continue
        agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, v in agg.items():
    print("%-28s n=%d  mean per dispatch %.4g" % (k, len(v), sum(v)/len(v)))
PY
