"""One wgrad shape, a few launches: target of tools/pmc_kernel.sh (rocprofv3 --pmc passes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flex-dm_amd"))
import torch
from mfp.hip import ops
K = 32768
M, N = int(os.environ.get("M", 256)), int(os.environ.get("N", 512))
A = torch.randn(K, M, device="cuda").bfloat16(); B = torch.randn(K, N, device="cuda").bfloat16()
out = torch.empty(M, N, device="cuda")
sk = ops.wgrad_splitk(K, M, N)
for _ in range(10):
    ops.gemm(A, B, M, N, K, a_kmajor=False, b_kmajor=False, out=out, splitk=sk)
torch.cuda.synchronize()
