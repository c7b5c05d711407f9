"""Stand-alone timing of the MX fp8 Dense (c5 Q|K|V shape) next to the bf16 product it replaces."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flex-dm_amd")]
import torch
from mfp.hip import ops
M, N, K = int(os.environ.get("M", 16384)), int(os.environ.get("N", 1536)), int(os.environ.get("K", 512))
dev = "cuda"
X = torch.randn(M, K, device=dev).to(torch.bfloat16)
W = torch.randn(N, K, device=dev) * 0.05
bias = torch.zeros(N, device=dev)
Wq = torch.empty(N * K, dtype=torch.uint8, device=dev); Ws = torch.empty(N * K // 32, dtype=torch.uint8, device=dev)
ops.quantize_mxfp8(W.reshape(-1), N, K, Wq, Ws)
Wb = W.to(torch.bfloat16)
def t(fn, n=20):
    for _ in range(3): fn()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
print("%s mx fp8 %d x %d x %d: %.1f us; bf16 mfp_gemm: %.1f us" % (os.environ.get("MFP_HIP_LIB", "default")[-24:], M, N, K,
      t(lambda: ops.gemm_mxfp8(X, Wq.view(N, K), Ws.view(N, K // 32), M, N, K, bias=bias)),
      t(lambda: ops.gemm(X, Wb, M, N, K, a_kmajor=True, b_kmajor=True, bias=bias, out_dtype=torch.bfloat16))))
