import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flex-dm_amd"))
import torch
from mfp.hip import ops
M, N, K = 32768, 768, 256
A = torch.randn(M, K, device="cuda").bfloat16(); B = torch.randn(N, K, device="cuda").bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(5):
    ops.gemm(A, B, M, N, K, a_kmajor=True, b_kmajor=True, out=out)
torch.cuda.synchronize()
