#!/usr/bin/env python
"""Condense a rocprofv3 results .db (--kernel-trace --stats) into a small CSV for profiles/."""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    if "at::native" in name[:60]:
        m = re.search(r"(vectorized_elementwise_kernel|elementwise_kernel_manual_unroll|distribution_elementwise_grid_stride_kernel|reduce_kernel|CatArrayBatchedCopy\w*|index\w*kernel\w*)", name)
        f = re.search(r"(\w+Functor\w*|direct_copy_kernel_cuda|uniform_kernel|normal_kernel|random_from_to_kernel|compare_scalar_kernel|where_kernel\w*)", name)
        return "torch:%s[%s]" % (m.group(1) if m else "kernel", f.group(1) if f else "?")
    return re.sub(r"\(.*$", "", name)[:90]


def main(db_path, out_path, steps):
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    agg = {}
    for name, calls, total, avg, pct in rows:
        k = short(name)
        a = agg.setdefault(k, [0, 0.0])
        a[0] += calls
        a[1] += total
    tot = sum(a[1] for a in agg.values())
    with open(out_path, "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats; durations in microseconds; %d train steps in the trace window\n" % steps)
        f.write("kernel,calls,total_us,avg_us,percent,us_per_step\n")
        for k, (calls, total) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            f.write("%s,%d,%.1f,%.2f,%.2f,%.1f\n" % (k.replace(",", ";"), calls, total, total / calls, 100 * total / tot, total / steps))
        f.write("# total_us,%.1f\n" % tot)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 1)
