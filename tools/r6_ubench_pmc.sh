#!/bin/bash
# Matrix-pipe / LDS counters of the chunk-loop microbenchmarks (the losing schedules beside the shipped one): rocprofv3 --pmc,
# kernel-trace only.  -> gpurun_out/r06_ubench_pmc.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/ubench_pmc
rm -rf $OUT; mkdir -p $OUT
for b in coresident pingpong; do
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/$b -o pmc -- \
    tools/ubench/$b > $OUT/$b.log 2>&1 || true
done
python - <<PY
import csv, glob, re, collections
out = open("gpurun_out/r06_ubench_pmc.txt", "w")
out.write("# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE over tools/ubench/{coresident,pingpong}\n")
out.write("# template arguments: coresident kc<ROWS, CHUNK COLUMNS, WAVES, WORKGROUPS PER CU>; pingpong k<MODE, STASH, DMA, PREFETCH> (modes: the file's header)\n")
out.write("# mfma_busy% = matrix-pipe busy cycles / (1024 SIMDs x kernel clocks); lds_busy% = LDS-array cycles / (256 CUs x kernel clocks)\n")
for b in ("coresident", "pingpong"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % b, recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(.*$", "", re.sub(r"^void ", "", r["Kernel_Name"])); agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE": cnt[k] += 1
    out.write("\n== %s\n%-40s %6s %14s %10s %9s %11s\n" % (b, "kernel", "calls", "clocks/launch", "mfma_busy%", "lds_busy%", "lds_confl%"))
    for k, c in sorted(agg.items()):
        clk = c["GRBM_GUI_ACTIVE"] / 8.0
        if clk <= 0: continue
        out.write("%-40s %6d %14.0f %10.1f %9.1f %11.1f\n" % (k, cnt[k], clk / cnt[k], 100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / clk,
                  100.0 * c["SQ_LDS_IDX_ACTIVE"] / 256.0 / clk, 100.0 * c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1.0)))
out.close()
print(open("gpurun_out/r06_ubench_pmc.txt").read())
PY
rm -rf $OUT/coresident $OUT/pingpong
