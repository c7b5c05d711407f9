"""Yardstick only (not on the product path): what the vendor GEMM library makes of the block's four weight-gradient
products dW = A^T B (bf16 operands [K][M], [K][N], K = tokens) -- the time a hand-written engine should be measured against."""
import os, sys, torch
T = int(os.environ.get("T", 32768)); D = int(os.environ.get("D", 256))
dev = "cuda"
rnd = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)
shapes = [(3 * D, D), (2 * D, D), (D, 2 * D), (D, D)]
ops = [(rnd(T, m), rnd(T, n)) for m, n in shapes]
def run():
    return [torch.mm(a.t(), b) for a, b in ops]
for _ in range(3): run()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(10): run()
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 100
byts = sum(T * (m + n) * 2 for m, n in shapes)
print("vendor library, 4 products T=%d D=%d: %.1f us per group (%.2f TB/s operands)" % (T, D, us, byts / us / 1e6))
for (m, n), (a, b) in zip(shapes, ops):
    for _ in range(3): torch.mm(a.t(), b)
    e0.record()
    for _ in range(20): torch.mm(a.t(), b)
    e1.record(); torch.cuda.synchronize()
    print("  %4d x %4d: %.1f us" % (m, n, e0.elapsed_time(e1) * 50))
