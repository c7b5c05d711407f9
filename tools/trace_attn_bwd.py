"""Phase timing of attn_block_bwd_kernel from the BB_ABL=9 build (tools/abl/build_abl.sh block_attn_bwd BB_ABL 9):
MFP_HIP_LIB=tools/abl/libmfp_block_attn_bwd_9.so python tools/trace_attn_bwd.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flex-dm_amd")]
import torch
from mfp.hip import ops
T = 32768; B = T // 128; dev = "cuda"
rnd = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)
qkv = (torch.randn(T, 768, device=dev) * 0.7).to(torch.bfloat16)
nvalid = torch.full((B,), 128, dtype=torch.int32, device=dev)
a, lse = ops.attention_fwd(qkv, nvalid, B, 128, 8)
d_o1, Wot, Wqt = rnd(T, 256), rnd(256, 256), rnd(256, 768)
for _ in range(3):
    dqkv, dy1 = ops.attn_block_bwd(d_o1, Wot, qkv, a, lse, nvalid, Wqt, B, 128, 8)
torch.cuda.synchronize()
tr = dy1.view(torch.int64).view(-1)[:B * 512].view(B, 8, 64).cpu().double()      # [doc][wave][64 stamps]       # [doc][wave][64 stamps]
t0 = tr[:, :, 0].min()
names = {0: "start", 1: "prologue issued", 2: "qkv placed (HBM latency)", 3: "chunk 0 landed", 4: "da product done"}
for pr in range(4):
    b = 5 + 12 * pr
    names.update({b: "p%d start" % pr, b + 1: "p%d frags+LsD" % pr, b + 2: "p%d attention done" % pr, b + 9: "p%d steps done" % pr, b + 10: "p%d next placed" % pr})
    for j in range(6):
        names[b + 3 + j] = "p%d step %d wait" % (pr, j)
names[53] = "dy1 image"
rel = tr - tr[:, :, :1].min(dim=1, keepdim=True).values      # from the workgroup's first wave start
clk = float(os.environ.get("CLK_MHZ", 100.0))     # s_memtime ticks per microsecond
prev = 0.0
for i in sorted(names):
    if i == 51:
        continue
    if i > 45 and (i - 5) % 12 == 10:
        continue
    v = rel[:, 0, i].mean().item() / clk
    print("%2d %-28s %8.2f us   +%6.2f   (min %.2f max %.2f over documents; wave spread %.2f)" % (
        i, names[i], v, v - prev, rel[:, 0, i].min().item() / clk, rel[:, 0, i].max().item() / clk,
        (rel[:, :, i].max(dim=1).values - rel[:, :, i].min(dim=1).values).mean().item() / clk))
    prev = v
print("start skew over documents: %.2f us" % ((tr[:, 0, 0].max() - t0).item() / clk))
