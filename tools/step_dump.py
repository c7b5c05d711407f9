"""List the kernels of one captured train step (rocprofv3 --kernel-trace DB): start offset, duration,
queue, name -- to read overlap between the main chain and the side-stream weight gradients."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, queue_id, start, end from kernels order by start").fetchall()
tc = [r[2] for r in rows if 'transpose_cast' in r[0]]
def short(n): return re.sub(r"^void ", "", re.sub(r"\(anonymous namespace\)::", "", n))[:60]
si = len(tc) // 2
s0, s1 = tc[si], tc[si + 1]
step = [r for r in rows if s0 <= r[2] < s1]
qs = sorted({r[1] for r in step})
print("step wall %.1f us, %d kernels, queues %s" % ((s1 - s0) / 1e3, len(step), qs))
busy_end = s0
for n, q, a, b in step:
    print("%8.1f %7.1f q%-2d %s" % ((a - s0) / 1e3, (b - a) / 1e3, qs.index(q), short(n)))
