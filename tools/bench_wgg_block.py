"""Stand-alone timing of the block's grouped weight-gradient launch (graph replay of 10)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flex-dm_amd")]
import torch
from mfp.hip import ops
T, D = int(os.environ.get("T", 32768)), 256
dev = "cuda"
rnd = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)
f = lambda *s: torch.empty(*s, device=dev)
dqkv, y1, dh, y2, d_o2, h, d_o1, a = rnd(T, 3 * D), rnd(T, D), rnd(T, 2 * D), rnd(T, D), rnd(T, D), rnd(T, 2 * D), rnd(T, D), rnd(T, D)
jobs = [dict(A=dqkv, B=y1, out=f(3 * D, D), M=3 * D, N=D, colsum=f(3 * D)), dict(A=dh, B=y2, out=f(2 * D, D), M=2 * D, N=D, colsum=f(2 * D)),
        dict(A=d_o2, B=h, out=f(D, 2 * D), M=D, N=2 * D), dict(A=d_o1, B=a, out=f(D, D), M=D, N=D)]
fn = lambda: ops.wgrad_group(jobs, T, None)
for _ in range(3): fn()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(10): fn()
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 100
byts = sum(T * (j["M"] + j["N"]) * 2 for j in jobs)
print("wgg block group %-28s %.1f us  (%.2f TB/s operands)" % (os.environ.get("MFP_HIP_LIB", "default")[-26:], us, byts / us / 1e6))
