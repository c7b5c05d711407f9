#!/bin/bash
# Matrix-pipe and LDS utilisation per kernel of the train step (rocprofv3 --pmc, kernel-trace only):
#   SQ_VALU_MFMA_BUSY_CYCLES (cycles the matrix pipe is busy, summed over the chip's 1024 SIMDs),
#   SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE (kernel duration in shader clocks, summed over 8 XCDs),
#   SQ_LDS_IDX_ACTIVE / SQ_LDS_BANK_CONFLICT (LDS-array cycles, summed over 256 CUs).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_mfma
rm -rf $OUT; mkdir -p $OUT
STEPS=${STEPS:-6}
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/p -o pmc -- \
  python bench.py --steps $STEPS --warmup 1 --no-cpu-baseline --no-roofline > $OUT/run.log 2>&1 || true
python - <<PY
import csv, glob, re, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    return "torch:elementwise/reduce" if "at::native" in n[:80] else re.sub(r"\(.*$", "", n)[:64]
for f in glob.glob("$OUT/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"]); agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE": cnt[k] += 1
rows = []
for k, c in agg.items():
    clk = c["GRBM_GUI_ACTIVE"] / 8.0                      # per-XCD sum -> shader clocks of the launches
    if clk <= 0: continue
    rows.append((clk, k, cnt[k], 100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / clk, 100.0 * c["SQ_LDS_IDX_ACTIVE"] / 256.0 / clk,
                 100.0 * c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1.0)))
tot = sum(r[0] for r in rows)
with open("gpurun_out/${TAG:-r03}_pmc_mfma_lds.txt", "w") as f:
    f.write("# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE over bench.py (c2, bf16)\n")
    f.write("# mfma_busy%% = matrix-pipe busy cycles / (1024 SIMDs x kernel clocks); lds_busy%% = LDS-array cycles / (256 CUs x kernel clocks)\n")
    f.write("%-66s %7s %9s %10s %9s %11s\n" % ("kernel", "calls", "time%", "mfma_busy%", "lds_busy%", "lds_confl%"))
    for clk, k, n, m, l, b in sorted(rows, reverse=True):
        f.write("%-66s %7d %9.2f %10.1f %9.1f %11.1f\n" % (k, n, 100 * clk / tot, m, l, b))
    wm = sum(r[0] * r[3] for r in rows) / tot
    f.write("# time-weighted matrix-pipe busy over the whole step: %.1f %%\n" % wm)
print(open("gpurun_out/${TAG:-r03}_pmc_mfma_lds.txt").read())
PY
