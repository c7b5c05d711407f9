import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flex-dm_amd"))
import torch
from mfp.hip import ops
B, S, H, hd = 256, 128, 8, 32
qkv = torch.randn(B * S, 3 * H * hd, device="cuda").bfloat16()
dout = torch.randn(B * S, H * hd, device="cuda").bfloat16()
nv = torch.full((B,), S, dtype=torch.int32, device="cuda")
for _ in range(4):
    out, lse = ops.attention_fwd(qkv, nv, B, S, H)
    dqkv = ops.attention_bwd(qkv, nv, out, dout, lse, B, S, H)
torch.cuda.synchronize()
