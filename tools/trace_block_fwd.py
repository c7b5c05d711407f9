"""Phase timing of the whole-block forward kernel from the AB_ABL=9 build (tools/abl/build_abl.sh block_attn AB_ABL 9):
MFP_HIP_LIB=tools/abl/libmfp_block_attn_9.so CLK_MHZ=2080 python tools/trace_block_fwd.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flex-dm_amd")]
import torch
from mfp.hip import ops
T, D = int(os.environ.get("T", 32768)), 256; B = T // 128; dev = "cuda"
HALF = int(os.environ.get("HALF", "0"))      # mfp_block_fwd_xhat_half: two workgroups per document, HALF = 4 / 8 waves each (1 = 4)
HALF = 4 if HALF == 1 else HALF
rnd = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)
x = torch.randn(T, D, device=dev)
gam, bet = torch.rand(D, device=dev) + 0.5, torch.randn(D, device=dev)
Wq, bq = rnd(768, 256), torch.randn(768, device=dev)
Wo, bo = rnd(256, 256), torch.randn(256, device=dev)
W1, b1, W2, b2 = rnd(512, 256), torch.randn(512, device=dev), rnd(256, 512), torch.randn(256, device=dev)
g2, be2 = torch.rand(D, device=dev) + 0.5, torch.randn(D, device=dev)
nvalid = torch.full((B,), 128, dtype=torch.int32, device=dev)
step = torch.zeros(1, dtype=torch.int32, device=dev)
x2c = torch.zeros(T, D, dtype=torch.bfloat16, device=dev)
for _ in range(3):
    ops.block_fwd(x, gam, bet, Wq, bq, Wo, bo, nvalid, g2, be2, W1, b1, W2, b2, B, 128, 8, 0.1, 5, 3, 4, step, x2_c=x2c,
                  xhat_stash=True, half_tiles=HALF)
torch.cuda.synchronize()
tr = x2c.view(torch.int64).view(-1)[:B * 512 * (2 if HALF == 8 else 1)].view((2 * B, HALF, 64) if HALF else (B, 8, 64)).cpu().double()
rel = tr - tr[:, :, :1].min(dim=1, keepdim=True).values
names = {0: "start", 1: "x landed", 2: "LN1 done, y1 out"}
for c in range(16):
    names[3 + c] = "chunk %2d (pair %d %s)" % (c, c >> 2, "qkv+attn,o"[0:0] + ["q", "k", "v + attention", "o"][c & 3])
names.update({19: "x1 epilogue", 20: "LN2 statistics", 21: "y2 out, fragments"})
for cm in range(16):
    names[22 + cm] = "mlp chunk %2d (%s)" % (cm, ["ffn1 a", "ffn1 b", "ffn2 a", "ffn2 b"][cm & 3])
clk = float(os.environ.get("CLK_MHZ", 2080.0))
prev = 0.0
for i in sorted(names):
    v = rel[:, 0, i].mean().item() / clk
    print("%2d %-30s %8.2f us   +%6.2f   (min %.2f max %.2f over documents)" % (i, names[i], v, v - prev, rel[:, 0, i].min().item() / clk, rel[:, 0, i].max().item() / clk))
    prev = v
