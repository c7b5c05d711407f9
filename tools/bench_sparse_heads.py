"""Event timing of the pieces of the compacted numerical-head path at the timed shape (T = 32768, 15 % of the rows)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flex-dm_amd")]
import torch
from mfp.hip import ops

dev = "cuda"
T, D, N = 32768, 256, 512
n = int(0.15 * T)
cnt = torch.tensor([n], dtype=torch.int32, device=dev)
idx = torch.randperm(T, device=dev)[:T].sort().values.to(torch.int32)
X = torch.randn(T, D, device=dev).to(torch.bfloat16)
W = (torch.randn(N, D, device=dev) * 0.05).to(torch.bfloat16)
dY = torch.randn(T, N, device=dev).to(torch.bfloat16)
out = torch.empty(T, N, device=dev)
dx = torch.empty(T, D, device=dev)
dst = torch.randn(T, D, device=dev)
xc = torch.empty_like(X)
gW, gb = torch.empty(N, D, device=dev), torch.empty(N, device=dev)


def timeit(name, fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    print("%-34s %7.1f us" % (name, 1e3 * e0.elapsed_time(e1) / reps))


with torch.cuda.stream(torch.cuda.Stream()):
    timeit("gather_rows", lambda: ops.gather_rows(X, idx, cnt, out=xc))
    timeit("gemm ws m_dev (fwd pred)", lambda: ops.gemm(xc, W, T, N, D, a_kmajor=True, b_kmajor=True, out=out, m_dev=cnt))
    timeit("gemm ws full T", lambda: ops.gemm(xc, W, T, N, D, a_kmajor=True, b_kmajor=True, out=out))
    timeit("gemm tiled m_dev (dgrad)", lambda: ops.gemm(dY, W, T, D, N, a_kmajor=True, b_kmajor=False, out=dx, m_dev=cnt))
    timeit("gemm tiled full T", lambda: ops.gemm(dY, W, T, D, N, a_kmajor=True, b_kmajor=False, out=dx))
    timeit("scatter_add_rows", lambda: ops.scatter_add_rows(dst, dx, idx, cnt))
    for sk in (8, 16, 32):
        timeit("wgrad k_dev x2 jobs splitk %d" % sk, lambda: ops.wgrad_group(
            [dict(A=dY, B=xc, out=gW, M=N, N=D, colsum=gb, k_dev=cnt), dict(A=dY, B=xc, out=gW, M=N, N=D, colsum=gb, k_dev=cnt)], T, splitk=sk))
