"""Main-path vs side-path time of one captured train step from a rocprofv3 --kernel-trace DB
(side = weight-gradient products and their reductions; everything else is the dependent main chain)."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, queue_id, start, end from kernels order by start").fetchall()
tc = [r[2] for r in rows if 'transpose_cast' in r[0]]
side = lambda n: ('gemm_wg_kernel' in n) or ('gemm_kernel<unsigned short, false, false' in n) or ('splitk_reduce' in n) or ('embed_onehot' in n)
def short(n): return re.sub(r"^void ", "", re.sub(r"\(anonymous namespace\)::", "", n))[:44]
for si in (len(tc) // 2,):
    s0, s1 = tc[si], tc[si + 1]
    step = [r for r in rows if s0 <= r[2] < s1]
    main = sorted([r for r in step if not side(r[0])], key=lambda r: r[2]); sd = [r for r in step if side(r[0])]
    mb = sum(r[3] - r[2] for r in main) / 1e3; sb = sum(r[3] - r[2] for r in sd) / 1e3
    print("step wall %.0f us; main kernels %d sum %.0f us; side kernels %d sum %.0f us" % ((s1 - s0) / 1e3, len(main), mb, len(sd), sb))
    tot = 0
    for i in range(len(main) - 1):
        g = (main[i + 1][2] - main[i][3]) / 1e3
        if g > 6:
            tot += g
            print("   main idle %6.1f us at %7.1f after %-44s before %s" % (g, (main[i][3] - s0) / 1e3, short(main[i][0]), short(main[i + 1][0])))
    print("   sum of main idle periods > 6 us: %.0f us" % tot)
