#!/usr/bin/env python
"""Stand-alone timing of the bf16 attention kernels at the timed shape (B=256, S=128, 8 heads of 32)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flex-dm_amd")]
import torch  # noqa: E402

from mfp.hip import ops  # noqa: E402

B, S, H, hd = int(os.environ.get("B", 256)), int(os.environ.get("S", 128)), 8, int(os.environ.get("HD", 32))
D = H * hd
dev = "cuda"


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


qkv = torch.randn(B * S, 3 * D, device=dev).to(torch.bfloat16)
nvalid = torch.full((B,), S, dtype=torch.int32, device=dev)
o, lse = ops.attention_fwd(qkv, nvalid, B, S, H)
do = torch.randn(B * S, D, device=dev).to(torch.bfloat16)
tf = timeit(lambda: ops.attention_fwd(qkv, nvalid, B, S, H))
tb = timeit(lambda: ops.attention_bwd(qkv, nvalid, o, do, lse, B, S, H))
nb_f, nb_b = B * S * 4 * D * 2, B * S * 8 * D * 2
print("attention fwd %.1f us (%.2f TB/s)   bwd %.1f us (%.2f TB/s)   [two-pass env: %s]"
      % (tf, nb_f / tf / 1e6, tb, nb_b / tb / 1e6, os.environ.get("MFP_ATTN_BWD_TWO_PASS")))
